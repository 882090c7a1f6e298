#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
echo "== abl"; timeout 100 scripts/lab/_bin/gemm_lab abl > gpurun_out/r03/lab_split_abl.txt 2>&1
echo "== tests"; timeout 1200 python -m pytest tests/test_gemnet_gpu.py tests/test_escn_gpu.py tests/test_equiformer_gpu.py tests/test_qhnet_gpu.py tests/test_phisnet_gpu.py tests/test_schnet_gpu.py tests/test_mixed_sizes_gpu.py tests/test_graphed_gpu.py -q -m gpu --tb=short 2>&1 | tail -25
