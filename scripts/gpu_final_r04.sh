#!/bin/bash
# Round-4 evidence run on the GPU box (hard timeouts on every step).  Everything lands in gpurun_out/r04_final/; summaries are copied to profiles/ afterwards.
#   1 full GPU test suite + smoke   2 the driver's bench command (default)   3 bench --full   4 rocprofv3 kernel stats of the default command
#   5 PMC passes FETCH_SIZE / WRITE_SIZE (separate runs, kernel trace only) for PaiNN B = 2048 and QHNet B = 16   6 QHNet record + kernel stats
OUT=gpurun_out/r04_final
mkdir -p $OUT
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.txt gpurun_out/mixed_sizes_report.txt
echo "== tests";  timeout -k 5 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee $OUT/pytest_gpu.log
cp gpurun_out/parity_report.txt gpurun_out/mixed_sizes_report.txt $OUT/ 2>/dev/null
echo "== smoke";  timeout -k 5 400 python __graft_entry__.py --smoke 2>&1 | tail -8 | tee $OUT/smoke.log
echo "== bench default"; S=$(date +%s); timeout -k 5 600 python bench.py > $OUT/bench_default.stdout 2> $OUT/bench_default.err; echo "wall $(( $(date +%s) - S )) s" | tee $OUT/bench_default.wall
tail -1 $OUT/bench_default.stdout > $OUT/bench_default.json; wc -c $OUT/bench_default.json
cp gpurun_out/bench_full.json $OUT/bench_default_full_record.json; cp gpurun_out/kernel_events.txt $OUT/kernel_events_b2048.txt
echo "== bench qhnet"; timeout -k 5 400 python bench.py --model qhnet > $OUT/bench_qhnet.stdout 2> $OUT/bench_qhnet.err; tail -1 $OUT/bench_qhnet.stdout > $OUT/bench_qhnet.json; cp gpurun_out/bench_full.json $OUT/bench_qhnet_full_record.json
prof() {  # name, command...
  local name=$1; shift
  rm -rf $OUT/prof_$name
  timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o $name -- "$@" > $OUT/rocprof_$name.log 2>&1
  f=$(find $OUT/prof_$name -name "*kernel_stats*.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/${name}_kernel_stats.csv && head -5 "$f"
  rm -rf $OUT/prof_$name
}
echo "== rocprof painn";  prof painn_b2048 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline
echo "== rocprof qhnet";  prof qhnet_b16 python scripts/bench_qhnet.py --molecules 16 --steps 5 --warmup 2
pmc() {  # tag, batch-key, dest, command...
  local tag=$1 key=$2 dest=$3; shift 3
  rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
  timeout -k 5 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o p -- "$@" > $OUT/pmc_fetch_$tag.log 2>&1
  timeout -k 5 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o p -- "$@" > $OUT/pmc_write_$tag.log 2>&1
  python scripts/pmc_summary.py $key "$*" $dest | head -8
  rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
}
echo "== pmc painn"; pmc painn 2048 $OUT/pmc_traffic_painn.json python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
echo "== pmc qhnet"; pmc qhnet 16 $OUT/pmc_traffic_qhnet.json python scripts/bench_qhnet.py --molecules 16 --steps 2 --warmup 1
echo "== bench --full"; S=$(date +%s); timeout -k 5 900 python bench.py --full > $OUT/bench_full.stdout 2> $OUT/bench_full.err; echo "wall $(( $(date +%s) - S )) s" | tee $OUT/bench_full.wall
cp gpurun_out/bench_full.json $OUT/bench_full_record.json
ls -la $OUT
