#!/bin/bash
# Where does the per-molecule rbf_proj gradient (csrc/molpair.hip) start to pay?  Step time with the pair-row path forced (NQ_NO_MOLGW=1) and the per-molecule
# path forced (NQ_MOLGW=1) at 64 .. 512 conformers per step (the default switches at 4096 atoms ~ 98 conformers).
for b in 64 128 256 512; do
  for mode in NQ_NO_MOLGW NQ_MOLGW NQ_NO_MOLGW NQ_MOLGW; do
    v=$(env $mode=1 timeout 120 python bench.py --batch $b --steps 30 --warmup 5 --sustain 0 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'], 3))")
    echo "batch $b $mode=1  ms_per_step $v"
  done
done
