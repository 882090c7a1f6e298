#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_plugin_surface.py -m gpu -q --tb=short -k "golden or fused or plugin or lightning or energy_only" 2>&1 | tail -12
for mode in sorted mol; do
  echo "== NQ_GWR=$mode"
  NQ_GWR=$mode timeout 300 python bench.py --steps 5 --warmup 2 --batch 1024 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo
  head -9 gpurun_out/kernel_events.txt | tail -8
done
