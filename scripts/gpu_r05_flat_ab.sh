#!/bin/bash
# FlatParameters gradient gather on / off on ONE box, alternating: step times of the autograd-driven models
for rep in 1 2; do
for g in 0 1; do
  echo "== NQ_FLAT_GATHER=$g (rep $rep)"
  export NQ_FLAT_GATHER=$g
  timeout 400 python scripts/bench_phisnet.py 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('phisnet eager', {k: round(d[k], 2) for k in ('wall_ms_per_step', 'ms_per_step')})"
  timeout 400 python scripts/bench_phisnet.py --graph 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('phisnet graph', {k: round(d[k], 2) for k in ('wall_ms_per_step', 'ms_per_step')})"
  for m in qhnet equiformer; do
    timeout 400 python scripts/bench_graphed.py --model $m 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$m', {k: round(v, 3) for k, v in d.items() if k.endswith('ms_per_step')})"
  done
  timeout 300 python scripts/bench_qhnet.py --molecules 16 --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('qhnet16', round(d['ms_per_step'], 3))"
done
done
