#!/bin/bash
# Round 5: quick loop -- engine parity subset + default bench with the per-kernel HIP-event table
OUT=gpurun_out/r05_b; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 5 600 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "golden or deterministic or every_row or direct or dense" 2>&1 | tail -5 | tee $OUT/pytest_engine.log
timeout -k 5 300 python bench.py --no-cpu-baseline > $OUT/bench_mol.stdout 2> $OUT/bench_mol.err; tail -1 $OUT/bench_mol.stdout > $OUT/bench_mol.json; cp gpurun_out/kernel_events.txt $OUT/kernel_events_mol.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_b/bench_mol.json'))
print(d['value'], d['ms_per_step'], d.get('reference_batch_size_32'))
PY
head -12 $OUT/kernel_events_mol.txt
