#!/bin/bash
OUT=gpurun_out/r03
mkdir -p $OUT
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.txt
echo "== parity tests (report only)"; NQ_PARITY_REPORT_ONLY=1 timeout 900 python -m pytest tests/test_escn_gpu.py tests/test_equiformer_gpu.py tests/test_gemnet_gpu.py -q -m gpu --tb=short 2>&1 | tail -15
cat gpurun_out/parity_report.txt
echo "== smoke"; timeout 600 python __graft_entry__.py --smoke 2>&1 | grep -v "amdgpu.ids" | tail -12
echo "== 8-rank rehearsal"; timeout 1500 python -m pytest tests/test_dist_gpu.py -q -m gpu --tb=short -k rehearsal 2>&1 | tail -25
