#!/bin/bash
# split-bf16 GEMM engine: lab table (lib vs exact-f32 engine), GEMM tests, model steps with both engines
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
echo "== lab"; timeout 200 scripts/lab/_bin/gemm_lab split > gpurun_out/r03/lab_split8.txt 2>&1; grep -c "err" gpurun_out/r03/lab_split8.txt
echo "== tests"; timeout 600 python -m pytest tests/test_engine_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -15
for m in gemnet escn equiformer qhnet; do
  for f32 in 0 1; do
  NQ_GEMM_F32=$f32 timeout 600 python scripts/bench_$m.py --molecules 16 --steps 4 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$m f32=$f32', round(d['ms_per_step'],2), 'ms/step', 'loss', d.get('final_loss'))"
  done
done
for f32 in 0 1; do NQ_GEMM_F32=$f32 timeout 600 python bench.py --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('painn f32=$f32', d['value'], d['ms_per_step'])"; done
