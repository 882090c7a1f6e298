#!/bin/bash
# Development: build libnablaq variants with one cost removed from the dual-reverse message kernel (NQ_ABLATE in edge.hip)
# and time them.  Results are WRONG by construction -- timing only.
#   scripts/ablate.sh build    (container)      scripts/ablate.sh run   (GPU box)
set -e
cd "$(dirname "$0")/.."
D=nabladft_amd/_ablate
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=on"
if [ "$1" = build ]; then
  mkdir -p $D
  for k in ${ABLATIONS:-1 2 3 4}; do
    X="-DNQ_ABLATE=$k"
    [ $k = 7 ] && X="-DNQ_DUAL2_THREADS=1024"     # dual reverse with 16 waves per CU
    [ $k = 8 ] && X="-DNQ_DUAL2_THREADS=512 -DNQ_TAN2_THREADS=768"
    [ $k = 9 ] && X="-DNQ_DUAL2_THREADS=512 -DNQ_TAN2_THREADS=512"
    [ $k = 10 ] && X="-DNQ_CH_FWD=1 -DNQ_CH_TAN=1 -DNQ_CH_FORCE=1 -DNQ_CH_DUAL=1"   # two 64-channel slices per atom at F=128
    [ $k = 11 ] && X="-DNQ_CH_FWD=1"
    [ $k = 12 ] && X="-DNQ_CLAIM_KINDS=0"    # static row striding everywhere
    [ $k = 13 ] && X="-DNQ_CLAIM_KINDS=15 -DNQ_CLAIM_ROWS=4"
    [ $k = 14 ] && X="-DNQ_CLAIM_KINDS=8 -DNQ_CLAIM_ROWS=2"
    [ $k = 15 ] && X="-DNQ_CLAIM_KINDS=8 -DNQ_CLAIM_ROWS=1"
    [ $k = 16 ] && X="-DNQ_CLAIM_KINDS=15 -DNQ_CLAIM_ROWS=8"
    [ $k = 17 ] && X="-DNQ_CLAIM_KINDS=15 -DNQ_CLAIM_GROUPS=8"
    [ $k = 18 ] && X="-DNQ_CLAIM_KINDS=15 -DNQ_CLAIM_GROUPS=4"
    [ $k = 19 ] && X="-DNQ_CLAIM_KINDS=15 -DNQ_CLAIM_GROUPS=2"
    [ $k = 20 ] && X="-DNQ_CLAIM_GROUPS_DUAL=2"
    [ $k = 21 ] && X="-DNQ_CLAIM_GROUPS_DUAL=4 -DNQ_CLAIM_GROUPS=3"
    [ $k = 22 ] && X="-DNQ_CLAIM_GROUPS=2 -DNQ_CLAIM_ROWS=2"
    [ $k = 23 ] && X="-DNQ_DUAL_NA_RESIDENT=0 -DNQ_DUAL_KX_EARLY=1 -DNQ_DUAL_MIRROR=0"
    [ $k = 24 ] && X="-DNQ_DUAL_NA_RESIDENT=1 -DNQ_DUAL_KX_EARLY=1 -DNQ_DUAL_MIRROR=1"
    [ $k = 25 ] && X="-DNQ_DUAL_NA_RESIDENT=0 -DNQ_DUAL_KX_EARLY=1 -DNQ_DUAL_MIRROR=0 -DNQ_DUAL2_THREADS=384"
    /opt/rocm/bin/hipcc $FLAGS $X -c nabladft_amd/csrc/edge.hip -o $D/edge_$k.o &
  done
  wait
  for k in ${ABLATIONS:-1 2 3 4}; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libnablaq_$k.so nabladft_amd/csrc/_obj/graph.o nabladft_amd/csrc/_obj/gemm.o $D/edge_$k.o nabladft_amd/csrc/_obj/node.o nabladft_amd/csrc/_obj/schnet.o nabladft_amd/csrc/_obj/hblock.o nabladft_amd/csrc/_obj/so3.o nabladft_amd/csrc/_obj/geobasis.o nabladft_amd/csrc/_obj/engine.o
  done
  ls -la $D/*.so
else
  mkdir -p gpurun_out
  for k in 0 ${ABLATIONS:-1 2 3 4}; do
    lib=$PWD/$D/libnablaq_$k.so; [ $k = 0 ] && lib=$PWD/nabladft_amd/libnablaq.so
    echo "== ablate $k"
    NABLAQ_LIB=$lib timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:v for k,v in d['kernel_ms_per_step'].items() if k.startswith('msgf') or k.startswith('gwr')}, d['ms_per_step'])"
  done 2>&1 | tee gpurun_out/ablate.log
fi
