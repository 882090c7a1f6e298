#!/bin/bash
# Round 5: the tests that did not run in r05_c (stopped at the first failure), then the default bench
OUT=gpurun_out/r05_d; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 5 1500 python -m pytest tests/test_graphed_gpu.py tests/test_hblock_gpu.py tests/test_mixed_sizes_gpu.py tests/test_phisnet_gpu.py tests/test_qhnet_gpu.py tests/test_rccl_gpu.py tests/test_schnet_gpu.py tests/test_so3_gpu.py tests/test_spk_gpu.py tests/test_split_engine_gpu.py -q -m gpu 2>&1 | tail -8 | tee $OUT/pytest_rest.log
timeout -k 5 400 python bench.py > $OUT/bench_default.stdout 2> $OUT/bench_default.err; tail -1 $OUT/bench_default.stdout > $OUT/bench_default.json; cp gpurun_out/kernel_events.txt $OUT/kernel_events.txt; cp gpurun_out/bench_full.json $OUT/bench_default_full_record.json
tail -c 1500 $OUT/bench_default.json; echo; head -8 $OUT/kernel_events.txt; tail -3 $OUT/bench_default.err
