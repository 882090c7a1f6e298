OUT=gpurun_out/r04_final; mkdir -p $OUT
S=$(date +%s); timeout -k 5 600 python bench.py > $OUT/bench_default.stdout 2> $OUT/bench_default.err; echo "wall $(( $(date +%s) - S )) s" > $OUT/bench_default.wall
tail -1 $OUT/bench_default.stdout > $OUT/bench_default.json; cp gpurun_out/bench_full.json $OUT/bench_default_full_record.json
timeout -k 5 400 python bench.py --model qhnet > $OUT/bench_qhnet.stdout 2> $OUT/bench_qhnet.err; tail -1 $OUT/bench_qhnet.stdout > $OUT/bench_qhnet.json; cp gpurun_out/bench_full.json $OUT/bench_qhnet_full_record.json
S=$(date +%s); timeout -k 5 900 python bench.py --full > $OUT/bench_full.stdout 2> $OUT/bench_full.err; echo "wall $(( $(date +%s) - S )) s" > $OUT/bench_full.wall
cp gpurun_out/bench_full.json $OUT/bench_full_record.json
cat $OUT/*.wall; tail -c 600 $OUT/bench_default.json
