#!/bin/bash
# Development: libnablaq variants (one -D set per variant, one object replaced), built in the container, timed on the GPU box.
#   scripts/variants.sh build FILE "name1:-DX=1" "name2:-DY=2 -DZ=3" ...     scripts/variants.sh run [bench args]
set -e; set +e
cd "$(dirname "$0")/.."
D=nabladft_amd/_variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=on ${VAR_BASE_FLAGS:-}"
OBJS="graph gemm gemm_bf16 edge molpair updfuse node schnet hblock so3 qhnet qhgen gemnet_graph gemnet escn equiformer geobasis rccl engine"
if [ "$1" = build ]; then
  mkdir -p $D; rm -f $D/*.so $D/*.o
  F=$2; shift 2
  for v in "$@"; do
    n=${v%%:*}; x=${v#*:}
    /opt/rocm/bin/hipcc $FLAGS $x -c nabladft_amd/csrc/$F.hip -o $D/${F}_$n.o &
  done
  wait
  for v in "$@"; do
    n=${v%%:*}
    L=""; for o in $OBJS; do if [ $o = $F ]; then L="$L $D/${F}_$n.o"; else L="$L nabladft_amd/csrc/_obj/$o.o"; fi; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libnablaq_$n.so $L
  done
  ls $D/*.so
else
  shift || true
  mkdir -p gpurun_out
  for lib in nabladft_amd/libnablaq.so nabladft_amd/_variants/libnablaq_*.so; do
    echo "== $lib"
    NABLAQ_LIB=$PWD/$lib timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:round(v,3) for k,v in d['kernel_ms_per_step'].items() if any(k.startswith(p) for p in '${VAR_KEYS:-msgf,gwr}'.split(','))}, round(d['ms_per_step'],3))"
  done
fi
