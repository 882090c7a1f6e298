#!/bin/bash
# Round-5 evidence, second part (after scripts/gpu_final_r05.sh on the same tree): the HBM PMC passes of the QHNet step on the round-5 build, then the two
# bench records again so that each cites the PMC file of THIS tree (bench.py reads profiles/r05_pmc_traffic*.json).
OUT=gpurun_out/r05_final; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python scripts/bench_qhnet.py --molecules 16 --steps 2 --warmup 1"
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout -k 5 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o p -- $CMD > $OUT/pmc_fetch_qhnet.log 2>&1
timeout -k 5 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o p -- $CMD > $OUT/pmc_write_qhnet.log 2>&1
python scripts/pmc_summary.py 16 "$CMD" $OUT/pmc_traffic_qhnet.json | head -10 | tee $OUT/pmc_traffic_qhnet.txt
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
cp $OUT/pmc_traffic_qhnet.json profiles/r05_pmc_traffic_qhnet.json
echo "== bench qhnet"; timeout -k 5 400 python bench.py --model qhnet > $OUT/bench_qhnet.stdout 2> $OUT/bench_qhnet.err; tail -1 $OUT/bench_qhnet.stdout > $OUT/bench_qhnet.json; cp gpurun_out/bench_full.json $OUT/bench_qhnet_full_record.json
echo "== bench default"; S=$(date +%s); timeout -k 5 500 python bench.py > $OUT/bench_default.stdout 2> $OUT/bench_default.err; echo "wall $(( $(date +%s) - S )) s" | tee $OUT/bench_default.wall
tail -1 $OUT/bench_default.stdout > $OUT/bench_default.json; cp gpurun_out/bench_full.json $OUT/bench_default_full_record.json; cp gpurun_out/kernel_events.txt $OUT/kernel_events_b2048.txt
tail -c 600 $OUT/bench_qhnet.json; echo; python - <<'PY'
import json
for f in ("bench_qhnet.json", "bench_default.json"):
    d = json.loads(open("gpurun_out/r05_final/" + f).read())
    r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], r.get("kernel"), r.get("frac"), r.get("traffic"), r.get("traffic_source"))
PY
