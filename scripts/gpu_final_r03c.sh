#!/bin/bash
# after the FlatParameters alignment: the default bench line again (its QHNet part changed) and QHNet's own record / kernel stats
OUT=gpurun_out/r03_final2
mkdir -p $OUT
export TMPDIR=/tmp
echo "== bench default"; S=$(date +%s); timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "wall $(( $(date +%s) - S )) s" | tee $OUT/bench_default.wall
cp gpurun_out/kernel_events.txt $OUT/ 2>/dev/null
echo "== bench qhnet"; timeout 600 python bench.py --model qhnet --steps 5 --warmup 2 > $OUT/bench_qhnet.json 2> $OUT/bench_qhnet.err
rm -rf $OUT/prof_q; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_q -o qhnet_b16 -- python scripts/bench_qhnet.py --molecules 16 --steps 5 --warmup 2 > $OUT/rocprof_qhnet_b16.log 2>&1
f=$(find $OUT/prof_q -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/qhnet_b16_kernel_stats.csv && head -5 "$f" | cut -c1-150; rm -rf $OUT/prof_q
echo "== qhnet tests"; timeout 600 python -m pytest tests/test_qhnet_gpu.py tests/test_phisnet_gpu.py -q -m gpu 2>&1 | tail -3
