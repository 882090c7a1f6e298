#!/bin/bash
# Round 5, first measurement of the molecule-per-workgroup rbf_proj gradient (csrc/molpair.hip) and the unpaired LDS filter taps:
# engine parity tests, then the default bench (per-kernel HIP-event table) with the new path and with the round-4 pair-row path.
OUT=gpurun_out/r05_a; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_engine_gpu.py tests/test_spk_gpu.py tests/test_schnet_gpu.py -x -q -m gpu 2>&1 | tail -15 | tee $OUT/pytest_engine.log
timeout -k 5 300 python bench.py --no-cpu-baseline > $OUT/bench_mol.stdout 2> $OUT/bench_mol.err; tail -1 $OUT/bench_mol.stdout > $OUT/bench_mol.json; cp gpurun_out/kernel_events.txt $OUT/kernel_events_mol.txt
NQ_NO_MOLGW=1 timeout -k 5 300 python bench.py --no-cpu-baseline > $OUT/bench_pairrows.stdout 2> $OUT/bench_pairrows.err; tail -1 $OUT/bench_pairrows.stdout > $OUT/bench_pairrows.json; cp gpurun_out/kernel_events.txt $OUT/kernel_events_pairrows.txt
tail -c 600 $OUT/bench_mol.json; echo; tail -c 300 $OUT/bench_pairrows.json; echo; head -40 $OUT/kernel_events_mol.txt
