#!/bin/bash
# Development: gemm micro-benchmark for alternative builds of gemm.hip (nabladft_amd/_ablate/libnablaq_g*.so)
for tag in base "$@"; do
  lib=$PWD/nabladft_amd/_ablate/libnablaq_$tag.so; [ $tag = base ] && lib=$PWD/nabladft_amd/libnablaq.so
  echo "== $tag"
  NABLAQ_LIB=$lib timeout 200 python scripts/gemm_bench.py 2>/dev/null | awk -F'|' '{print $1 "|" $3}'
done
