#!/bin/bash
# fused +-m pair products (nq_linear_forward_res): parity tests of the two models (default engines and split engine forced), benches
export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests/test_escn_gpu.py tests/test_equiformer_gpu.py tests/test_split_engine_gpu.py tests/test_mixed_sizes_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -12
for m in escn equiformer; do
  timeout 600 python scripts/bench_$m.py --molecules 16 --steps 4 --warmup 2 --kernels 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$m', round(d['ms_per_step'],2), 'ms/step', [(k,v[0],v[1]) for k,v in d['kernel_ms_per_step'].items() if k in ('gn_lincomb','gn_mul')], 'gemm ms', round(d['gemm_ms_per_step'],1))"
done
