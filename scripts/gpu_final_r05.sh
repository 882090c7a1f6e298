#!/bin/bash
# Round-5 evidence run on the GPU box (final tree): full GPU suite, smoke, the bench records (default = the driver's command, qhnet, --full), rocprofv3 kernel
# stats of the PaiNN and QHNet commands (kernel trace only), the two HBM PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, as MI355X_MICROARCH.md prescribes)
# and one L2-request pass.  Everything lands in gpurun_out/r05_final/; the summaries are copied to profiles/ afterwards.
OUT=gpurun_out/r05_final; mkdir -p $OUT; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.txt gpurun_out/mixed_sizes_report.txt
echo "== tests"; timeout -k 5 1500 python -m pytest tests -q -m gpu 2>&1 | tail -5 | tee $OUT/pytest_gpu.log
cp gpurun_out/parity_report.txt gpurun_out/mixed_sizes_report.txt $OUT/ 2>/dev/null
echo "== smoke"; timeout -k 5 400 python __graft_entry__.py --smoke 2>&1 | tail -7 | tee $OUT/smoke.log
echo "== bench default"; S=$(date +%s); timeout -k 5 500 python bench.py > $OUT/bench_default.stdout 2> $OUT/bench_default.err; echo "wall $(( $(date +%s) - S )) s" | tee $OUT/bench_default.wall
tail -1 $OUT/bench_default.stdout > $OUT/bench_default.json; cp gpurun_out/bench_full.json $OUT/bench_default_full_record.json; cp gpurun_out/kernel_events.txt $OUT/kernel_events_b2048.txt
echo "== bench qhnet"; timeout -k 5 400 python bench.py --model qhnet > $OUT/bench_qhnet.stdout 2> $OUT/bench_qhnet.err; tail -1 $OUT/bench_qhnet.stdout > $OUT/bench_qhnet.json; cp gpurun_out/bench_full.json $OUT/bench_qhnet_full_record.json
prof() {  # name, command...
  local name=$1; shift
  rm -rf $OUT/prof_$name
  timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o $name -- "$@" > $OUT/rocprof_$name.log 2>&1
  f=$(find $OUT/prof_$name -name "*kernel_stats*.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/${name}_kernel_stats.csv && head -8 "$f"
  rm -rf $OUT/prof_$name
}
echo "== rocprof painn"; prof painn_b2048 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline
echo "== rocprof qhnet"; prof qhnet_b16 python scripts/bench_qhnet.py --molecules 16 --steps 5 --warmup 2
echo "== pmc painn"
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
timeout -k 5 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o p -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout -k 5 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o p -- $CMD > $OUT/pmc_write.log 2>&1
python scripts/pmc_summary.py 2048 "$CMD" $OUT/pmc_traffic_painn.json | head -14 | tee $OUT/pmc_traffic_painn.txt
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
echo "== L2 requests"
rm -rf gpurun_out/pmc_tcc
timeout -k 5 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d gpurun_out/pmc_tcc -o p -- $CMD > $OUT/pmc_tcc.log 2>&1
python - <<'PY' | tee gpurun_out/r05_final/pmc_tcc_requests.txt
import csv, glob, collections
fs = glob.glob("gpurun_out/pmc_tcc/*counter_collection*.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
if fs:
    with open(fs[0]) as f:
        for row in csv.DictReader(f):
            k = row.get("Kernel_Name", "?").replace("void ", "").split("(")[0][:40]
            agg[k][row.get("Counter_Name")] += float(row.get("Counter_Value", 0)); cnt[(k, row.get("Counter_Name"))] += 1
names = sorted({c for v in agg.values() for c in v})
print("# rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline  (B = 2048, round-5 build)")
print(f"{'kernel':40s} " + " ".join(f"{n:>16s}" for n in names) + "   launches")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:12]:
    print(f"{k:40s} " + " ".join(f"{v[n] / max(cnt[(k, n)], 1):16.4g}" for n in names) + f"   {max(cnt[(k, n)] for n in names)}")
PY
rm -rf gpurun_out/pmc_tcc
echo "== bench --full"; S=$(date +%s); timeout -k 5 900 python bench.py --full > $OUT/bench_full.stdout 2> $OUT/bench_full.err; echo "wall $(( $(date +%s) - S )) s" | tee $OUT/bench_full.wall
cp gpurun_out/bench_full.json $OUT/bench_full_record.json
cat $OUT/*.wall; tail -c 400 $OUT/bench_default.json; echo; tail -c 300 $OUT/bench_qhnet.json; echo
