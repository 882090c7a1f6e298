#!/bin/bash
OUT=gpurun_out/r03
mkdir -p $OUT
export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests/test_escn_gpu.py tests/test_equiformer_gpu.py tests/test_mixed_sizes_gpu.py tests/test_graphed_gpu.py -q -m gpu --tb=short 2>&1 | tail -30 | tee $OUT/pytest_gpu_c.log
for m in escn equiformer; do
  echo "== bench $m"; timeout 600 python scripts/bench_$m.py --molecules 16 --steps 5 --warmup 2 --kernels > $OUT/bench_${m}_c.json 2> $OUT/bench_${m}_c.err; python - <<PY
import json
d=json.load(open("$OUT/bench_${m}_c.json"))
print("$m", round(d["ms_per_step"],2), "ms/step; gemm", round(d.get("gemm_ms_per_step",0),1), "ms; roofline frac", round(d["roofline"]["frac"],3))
print("   ", [(k,v[0],v[1]) for k,v in list(d["kernel_ms_per_step"].items())[:12]])
PY
done
