#!/bin/bash
# after the fused +-m pair products: eSCN / EquiformerV2 records and kernel stats again
OUT=gpurun_out/r03_final2
mkdir -p $OUT
export TMPDIR=/tmp
for m in escn equiformer; do
  echo "== bench $m"; timeout 600 python bench.py --model $m --steps 5 --warmup 2 > $OUT/bench_$m.json 2> $OUT/bench_$m.err
  rm -rf $OUT/prof_x; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_x -o ${m}_b16 -- python scripts/bench_$m.py --molecules 16 --steps 3 --warmup 1 > $OUT/rocprof_${m}_b16.log 2>&1
  f=$(find $OUT/prof_x -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${m}_b16_kernel_stats.csv && head -4 "$f" | cut -c1-150; rm -rf $OUT/prof_x
done
python - <<'P'
import json
for m in ("escn","equiformer"):
    x=json.loads(open(f"gpurun_out/r03_final2/bench_{m}.json").read().strip().splitlines()[-1])
    print(m, round(x["ms_per_step"],2), round(x["value"],2), x["roofline"]["frac"], x["roofline"]["achieved"], (x.get("bf16_mode") or {}).get("ms_per_step"), [v.get("ms_per_step") for k,v in x.items() if k.startswith("reference_batch_size") and isinstance(v,dict) and "ms_per_step" in v])
P
