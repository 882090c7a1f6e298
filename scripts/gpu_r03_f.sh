#!/bin/bash
export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests/test_gemnet_gpu.py tests/test_escn_gpu.py tests/test_equiformer_gpu.py tests/test_mixed_sizes_gpu.py tests/test_graphed_gpu.py -q -m gpu --tb=short 2>&1 | tail -8
for m in gemnet escn equiformer; do
  timeout 600 python scripts/bench_$m.py --molecules 16 --steps 4 --warmup 2 --kernels 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$m', round(d['ms_per_step'],2), 'ms/step', [(k,v[0],v[1]) for k,v in d['kernel_ms_per_step'].items() if k in ('gn_lincomb','gn_mul','s2act_fwd','s2act_bwd','rowop_fwd','rowop_tr','qh_act')])"
done
