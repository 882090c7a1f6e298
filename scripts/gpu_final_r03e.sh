#!/bin/bash
# last act of round 3: the full GPU suite and smoke on the final code
OUT=gpurun_out/r03_final2
mkdir -p $OUT
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.txt gpurun_out/mixed_sizes_report.txt
echo "== tests";  S=$(date +%s); timeout 600 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 | tee $OUT/pytest_gpu.log; echo "wall $(( $(date +%s) - S )) s"
cp gpurun_out/parity_report.txt gpurun_out/mixed_sizes_report.txt $OUT/ 2>/dev/null
echo "== smoke";  timeout 200 python __graft_entry__.py --smoke 2>&1 | tail -7 | tee $OUT/smoke.log
