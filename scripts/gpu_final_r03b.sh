#!/bin/bash
# Round-3 evidence run, second edition (after the split-bf16 GEMM engine): part A = full GPU test suite, smoke, the default bench line, rocprofv3 kernel stats of
# the same command; part B = the per-model bench lines and their rocprofv3 kernel stats.  Everything lands in gpurun_out/r03_final2/.
# usage: gpu_final_r03b.sh A|B
OUT=gpurun_out/r03_final2
mkdir -p $OUT
export TMPDIR=/tmp
prof() {  # name, command...
  local name=$1; shift
  rm -rf $OUT/prof_$name
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o $name -- "$@" > $OUT/rocprof_$name.log 2>&1
  f=$(find $OUT/prof_$name -name "*kernel_stats*.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/${name}_kernel_stats.csv && head -6 "$f" | cut -c1-160
  rm -rf $OUT/prof_$name
}
if [ "$1" = "A" ]; then
  rm -f gpurun_out/parity_report.txt gpurun_out/mixed_sizes_report.txt
  echo "== tests";  S=$(date +%s); timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee $OUT/pytest_gpu.log; echo "wall $(( $(date +%s) - S )) s"
  cp gpurun_out/parity_report.txt gpurun_out/mixed_sizes_report.txt $OUT/ 2>/dev/null
  echo "== smoke";  timeout 600 python __graft_entry__.py --smoke 2>&1 | tail -8 | tee $OUT/smoke.log
  echo "== bench default"; S=$(date +%s); timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "wall $(( $(date +%s) - S )) s" | tee $OUT/bench_default.wall
  cp gpurun_out/kernel_events.txt $OUT/ 2>/dev/null
  echo "== rocprof painn";  prof painn_b2048 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline
else
  for m in qhnet gemnet escn equiformer; do
    echo "== bench $m"; S=$(date +%s); timeout 600 python bench.py --model $m --steps 5 --warmup 2 > $OUT/bench_$m.json 2> $OUT/bench_$m.err; echo "wall $(( $(date +%s) - S )) s"
  done
  echo "== rocprof"; prof qhnet_b16 python scripts/bench_qhnet.py --molecules 16 --steps 5 --warmup 2
  prof gemnet_b16_f32 python scripts/bench_gemnet.py --molecules 16 --steps 5 --warmup 2
  prof escn_b16 python scripts/bench_escn.py --molecules 16 --steps 3 --warmup 1
  prof equiformer_b16 python scripts/bench_equiformer.py --molecules 16 --steps 3 --warmup 1
fi
ls $OUT | wc -l
