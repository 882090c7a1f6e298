#!/bin/bash
# Round 5: full GPU suite + smoke on the current tree
OUT=gpurun_out/r05_c; mkdir -p $OUT; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.txt gpurun_out/mixed_sizes_report.txt
timeout -k 5 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 | tee $OUT/pytest_gpu.log
cp gpurun_out/parity_report.txt gpurun_out/mixed_sizes_report.txt $OUT/ 2>/dev/null
timeout -k 5 400 python __graft_entry__.py --smoke 2>&1 | tail -7 | tee $OUT/smoke.log
