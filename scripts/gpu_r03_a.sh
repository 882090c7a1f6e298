#!/bin/bash
# round 3, first full check on the GPU box: whole GPU suite (new GEMM engine under every model, bf16 epoch fix, 10-90 atom mix) + per-model bench lines
OUT=gpurun_out/r03
mkdir -p $OUT
export TMPDIR=/tmp
echo "== tests"; timeout 1500 python -m pytest tests -q -m gpu --tb=short 2>&1 | tail -60 | tee $OUT/pytest_gpu_a.log
for m in qhnet gemnet escn equiformer; do
  echo "== bench $m"; timeout 600 python scripts/bench_$m.py --molecules 16 --steps 5 --warmup 2 --kernels > $OUT/bench_${m}_a.json 2> $OUT/bench_${m}_a.err; tail -c 600 $OUT/bench_${m}_a.json; echo
done
