#!/bin/bash
# QHNet step captured into a HIP graph after the detach fix: the bisect variants again, the replay == eager test, replay vs eager at 2 and 16 conformers
out=gpurun_out/r05_qhnet_capture_fixed.txt
: > $out
for w in only:node_embedding bwd_sum bwd_blocks bwd_loss; do
  echo "== $w" >> $out
  timeout 120 python scripts/debug_qhnet_capture.py $w 2>&1 | grep "OK\|require grad\|Fatal\|dumped" | head -4 >> $out
done
echo "== pytest" >> $out
timeout 600 python -m pytest tests/test_graphed_gpu.py -x -q -m gpu 2>&1 | tail -5 >> $out
for m in 2 16; do
  echo "== bench_graphed qhnet molecules=$m" >> $out
  timeout 300 python scripts/bench_graphed.py --model qhnet --molecules $m 2>&1 | grep -v amdgpu.ids | tail -3 >> $out
done
cat $out
