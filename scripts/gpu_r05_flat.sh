#!/bin/bash
# FlatParameters lazy gradient gather: tests that touch it, step times of the autograd-driven models, and the ATen launch counts of the QHNet step
OUT=gpurun_out/r05_flat; mkdir -p $OUT; export TMPDIR=/tmp
echo "== tests"; timeout -k 5 900 python -m pytest tests/test_graphed_gpu.py tests/test_phisnet_gpu.py tests/test_gemnet_gpu.py tests/test_rccl_gpu.py tests/test_dist_gpu.py tests/test_qhnet_gpu.py -x -q -m gpu 2>&1 | tail -4
for m in qhnet gemnet escn equiformer; do
  echo "== bench_graphed $m"; timeout 400 python scripts/bench_graphed.py --model $m 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: round(v, 3) for k, v in d.items() if k.endswith('ms_per_step')})"
done
echo "== qhnet 16"; timeout 300 python scripts/bench_qhnet.py --molecules 16 --steps 10 --warmup 3 2>&1 | tail -1 | cut -c1-300
echo "== phisnet"; timeout 400 python scripts/bench_phisnet.py 2>&1 | tail -2 | cut -c1-600
rm -rf $OUT/prof
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o qh -- python scripts/bench_qhnet.py --molecules 16 --steps 5 --warmup 2 > $OUT/rocprof.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/qhnet_b16_kernel_stats.csv; rm -rf $OUT/prof
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r05_flat/qhnet_b16_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
at = [r for r in rows if "at::native" in r["Name"] or "rocclr" in r["Name"]]
print("total ms", round(tot / 1e6, 2), "ATen ms", round(sum(float(r["TotalDurationNs"]) for r in at) / 1e6, 2), "ATen calls", sum(int(r["Calls"]) for r in at), "all calls", sum(int(r["Calls"]) for r in rows))
for r in sorted(at, key=lambda r: -float(r["TotalDurationNs"]))[:8]:
    print(round(float(r["TotalDurationNs"]) / 1e6, 2), r["Calls"], r["Name"][:140])
PY
