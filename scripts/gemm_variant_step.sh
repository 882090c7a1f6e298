#!/bin/bash
# Development: full training step with each GEMM kernel variant (bench.py --gemm-variant: bit0 8 waves, bit1 register prefetch, bit2 256x128 tiles)
for v in "$@"; do
  timeout 250 python bench.py --no-cpu-baseline --no-roofline --gemm-variant $v 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('variant $v', round(d['value']), round(d['ms_per_step'],2))
"
done
