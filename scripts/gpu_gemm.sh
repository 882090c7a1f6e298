#!/bin/bash
mkdir -p gpurun_out
timeout 600 python scripts/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gemm_bench.log
for v in 0 1 2 3; do echo "== bench variant $v"; timeout 300 python bench.py --steps 5 --warmup 2 --batch 1024 --no-cpu-baseline --gemm-variant $v 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo; done | tee gpurun_out/bench_variants.log
