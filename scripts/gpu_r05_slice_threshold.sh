#!/bin/bash
# Two 64-channel slices per CSR row (message kernels) below NQ_SMALL_SLICE_ATOMS atoms per launch: never (0), the shipped 3072, up to 8192 -- step time at 32 .. 192 conformers
for b in 32 64 96 128 192; do
  for lib in nabladft_amd/_variants/libnablaq_noslice.so nabladft_amd/libnablaq.so nabladft_amd/_variants/libnablaq_slice8k.so; do
    v=$(NABLAQ_LIB=$PWD/$lib timeout 120 python bench.py --batch $b --steps 40 --warmup 5 --sustain 0 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'], 3))")
    echo "batch $b $(basename $lib)  ms_per_step $v"
  done
done
