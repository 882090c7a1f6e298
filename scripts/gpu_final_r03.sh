#!/bin/bash
# Round-3 evidence run on the GPU box: full GPU test suite, smoke, the bench lines (default / qhnet / gemnet / escn / equiformer), rocprofv3 kernel stats of the same
# commands (kernel trace only) and the two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, as MI355X_MICROARCH.md prescribes).  Everything lands in
# gpurun_out/r03_final/; the summaries are copied to profiles/ afterwards.
OUT=gpurun_out/r03_final
mkdir -p $OUT
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.txt
echo "== tests";  timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee $OUT/pytest_gpu.log
cp gpurun_out/parity_report.txt $OUT/ 2>/dev/null
echo "== smoke";  timeout 600 python __graft_entry__.py --smoke 2>&1 | tail -8 | tee $OUT/smoke.log
echo "== bench default"; S=$(date +%s); timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "wall $(( $(date +%s) - S )) s" | tee $OUT/bench_default.wall
echo "== bench qhnet";   timeout 900 python bench.py --model qhnet > $OUT/bench_qhnet.json 2> $OUT/bench_qhnet.err
echo "== bench gemnet";  timeout 900 python bench.py --model gemnet > $OUT/bench_gemnet.json 2> $OUT/bench_gemnet.err
echo "== bench escn";    timeout 900 python bench.py --model escn --steps 5 --warmup 2 > $OUT/bench_escn.json 2> $OUT/bench_escn.err
echo "== bench equiformer"; timeout 900 python bench.py --model equiformer --steps 5 --warmup 2 > $OUT/bench_equiformer.json 2> $OUT/bench_equiformer.err
cp gpurun_out/kernel_events.txt $OUT/ 2>/dev/null
prof() {  # name, command...
  local name=$1; shift
  rm -rf $OUT/prof_$name
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o $name -- "$@" > $OUT/rocprof_$name.log 2>&1
  f=$(find $OUT/prof_$name -name "*kernel_stats*.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/${name}_kernel_stats.csv && head -8 "$f"
  rm -rf $OUT/prof_$name
}
echo "== rocprof painn";  prof painn_b2048 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline
echo "== rocprof qhnet";  prof qhnet_b16 python scripts/bench_qhnet.py --molecules 16 --steps 5 --warmup 2
echo "== rocprof gemnet"; prof gemnet_b16_f32 python scripts/bench_gemnet.py --molecules 16 --steps 5 --warmup 2
prof gemnet_b16_bf16 python scripts/bench_gemnet.py --molecules 16 --steps 5 --warmup 2 --precision bf16
prof escn_b16 python scripts/bench_escn.py --molecules 16 --steps 3 --warmup 1
prof equiformer_b16 python scripts/bench_equiformer.py --molecules 16 --steps 3 --warmup 1
echo "== pmc painn"
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o p -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o p -- $CMD > $OUT/pmc_write.log 2>&1
python scripts/pmc_summary.py 2048 "$CMD" $OUT/pmc_traffic_painn.json | head -8
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
echo "== graph replay at the reference batch sizes"
for m in qhnet gemnet escn equiformer; do timeout 300 python scripts/bench_graphed.py --model $m > $OUT/graphed_$m.json 2> $OUT/graphed_$m.err; tail -c 400 $OUT/graphed_$m.json; echo; done
echo "== gemm lab"; timeout 200 scripts/lab/_bin/gemm_lab > $OUT/gemm_lab.txt 2>&1
ls -la $OUT
