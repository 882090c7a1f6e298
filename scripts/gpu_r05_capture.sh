#!/bin/bash
# QHNet backward capture: which autograd nodes make hipStreamEndCapture crash?  (one process per variant)
out=gpurun_out/r05_qhnet_capture.txt
: > $out
for w in only:fc_ii.hamiltonian.2 only:fc_ii_bias only:fc_ij.hamiltonian.2 only:output_ii only:output_ij only:e3_gnn_node_pair_layer.1 only:e3_gnn_node_layer.1 only:e3_gnn_layer.4 only:e3_gnn_layer.0 only:node_embedding; do
  echo "== $w" >> $out
  timeout 120 python scripts/debug_qhnet_capture.py $w 2>&1 | grep -v "amdgpu.ids\|^Extension modules\|^$" | grep "OK\|require grad\|Error\|error\|dumped\|Fatal\|fault" | head -6 >> $out
done
echo "== AMD_LOG_LEVEL=3 bwd_sum (last HIP calls before the crash)" >> $out
AMD_LOG_LEVEL=3 timeout 300 python scripts/debug_qhnet_capture.py bwd_sum 2>&1 | grep -v "^Extension" | tail -120 | cut -c1-300 >> $out
tail -150 $out
