#!/bin/bash
OUT=gpurun_out/r03
mkdir -p $OUT
export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests/test_escn_gpu.py tests/test_equiformer_gpu.py tests/test_mixed_sizes_gpu.py -q -m gpu --tb=short 2>&1 | tail -25
for m in escn equiformer; do
  timeout 600 python scripts/bench_$m.py --molecules 16 --steps 4 --warmup 2 --kernels 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$m', round(d['ms_per_step'],2), 'ms/step; gemm', round(d.get('gemm_ms_per_step',0),1), 'ms; frac', round(d['roofline']['frac'],3))
print('   ', [(k,v[0],v[1]) for k,v in list(d['kernel_ms_per_step'].items())[:14]])"
done
