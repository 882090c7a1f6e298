#!/bin/bash
# Round-5 evidence, last edition (Python-side changes only since scripts/gpu_final_r05.sh + gpu_r05_evidence2.sh: FlatParameters gradient gather, stacked
# spherical-linear weights): full GPU suite, smoke, bench records (default, qhnet, --full) and the rocprofv3 kernel stats of the QHNet command (its launch mix
# changed).  The PaiNN rocprofv3 / PMC files of the earlier scripts stay valid: libnablaq.so is unchanged.
OUT=gpurun_out/r05_final; mkdir -p $OUT; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.txt gpurun_out/mixed_sizes_report.txt
echo "== tests"; timeout -k 5 1500 python -m pytest tests -q -m gpu 2>&1 | tail -5 | tee $OUT/pytest_gpu.log
cp gpurun_out/parity_report.txt gpurun_out/mixed_sizes_report.txt $OUT/ 2>/dev/null
echo "== smoke"; timeout -k 5 400 python __graft_entry__.py --smoke 2>&1 | tail -7 | tee $OUT/smoke.log
echo "== bench default"; S=$(date +%s); timeout -k 5 500 python bench.py > $OUT/bench_default.stdout 2> $OUT/bench_default.err; echo "wall $(( $(date +%s) - S )) s" | tee $OUT/bench_default.wall
tail -1 $OUT/bench_default.stdout > $OUT/bench_default.json; cp gpurun_out/bench_full.json $OUT/bench_default_full_record.json; cp gpurun_out/kernel_events.txt $OUT/kernel_events_b2048.txt
echo "== bench qhnet"; timeout -k 5 400 python bench.py --model qhnet > $OUT/bench_qhnet.stdout 2> $OUT/bench_qhnet.err; tail -1 $OUT/bench_qhnet.stdout > $OUT/bench_qhnet.json; cp gpurun_out/bench_full.json $OUT/bench_qhnet_full_record.json
echo "== rocprof qhnet"; rm -rf $OUT/prof_q
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_q -o qhnet_b16 -- python scripts/bench_qhnet.py --molecules 16 --steps 5 --warmup 2 > $OUT/rocprof_qhnet_b16.log 2>&1
f=$(find $OUT/prof_q -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/qhnet_b16_kernel_stats.csv && head -4 "$f"; rm -rf $OUT/prof_q
echo "== bench --full"; S=$(date +%s); timeout -k 5 900 python bench.py --full > $OUT/bench_full.stdout 2> $OUT/bench_full.err; echo "wall $(( $(date +%s) - S )) s" | tee $OUT/bench_full.wall
cp gpurun_out/bench_full.json $OUT/bench_full_record.json
tail -c 300 $OUT/bench_default.json; echo; tail -c 300 $OUT/bench_qhnet.json; echo
