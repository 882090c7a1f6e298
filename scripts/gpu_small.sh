#!/bin/bash
for b in 32 128 256 1024; do timeout 300 python bench.py --steps 20 --warmup 5 --batch $b --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo " batch=$b"; done
