#!/bin/bash
# round 3: graph replay at the reference batch sizes (one process per model: a failed capture must not poison the next), then the equality tests
OUT=gpurun_out/r03
mkdir -p $OUT
export TMPDIR=/tmp
for m in qhnet gemnet escn equiformer; do
  echo "== graphed $m"; timeout 300 python scripts/bench_graphed.py --model $m > $OUT/graphed_$m.json 2> $OUT/graphed_$m.err; echo rc=$?; tail -c 700 $OUT/graphed_$m.json; echo; grep -v "amdgpu.ids" $OUT/graphed_$m.err | tail -4
done
echo "== tests"; timeout 900 python -m pytest tests/test_graphed_gpu.py -q -m gpu --tb=short 2>&1 | tail -30 | tee $OUT/pytest_graphed.log
