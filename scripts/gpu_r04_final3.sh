#!/bin/bash
# Round-4 evidence, final edition (after the tensor-product tuning): GPU suite + smoke + the three bench records on the final code.
OUT=gpurun_out/r04_final3; mkdir -p $OUT; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.txt gpurun_out/mixed_sizes_report.txt
timeout -k 5 1000 python -m pytest tests -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest_gpu.log
cp gpurun_out/parity_report.txt gpurun_out/mixed_sizes_report.txt $OUT/ 2>/dev/null
timeout -k 5 300 python __graft_entry__.py --smoke 2>&1 | tail -7 | tee $OUT/smoke.log
S=$(date +%s); timeout -k 5 400 python bench.py > $OUT/bench_default.stdout 2> $OUT/bench_default.err; echo "wall $(( $(date +%s) - S )) s" > $OUT/bench_default.wall
tail -1 $OUT/bench_default.stdout > $OUT/bench_default.json; cp gpurun_out/bench_full.json $OUT/bench_default_full_record.json; cp gpurun_out/kernel_events.txt $OUT/kernel_events_b2048.txt
timeout -k 5 300 python bench.py --model qhnet > $OUT/bench_qhnet.stdout 2> $OUT/bench_qhnet.err; tail -1 $OUT/bench_qhnet.stdout > $OUT/bench_qhnet.json; cp gpurun_out/bench_full.json $OUT/bench_qhnet_full_record.json
S=$(date +%s); timeout -k 5 700 python bench.py --full > $OUT/bench_full.stdout 2> $OUT/bench_full.err; echo "wall $(( $(date +%s) - S )) s" > $OUT/bench_full.wall
cp gpurun_out/bench_full.json $OUT/bench_full_record.json
cat $OUT/*.wall; tail -c 300 $OUT/bench_default.json; echo; tail -c 400 $OUT/bench_qhnet.json
