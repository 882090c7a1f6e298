#!/bin/bash
# Development: build libnablaq variants with extra -D flags on ONE source file and time bench.py with each (NABLAQ_LIB).
#   scripts/variants.sh build <file.hip> name1="-DX=1 -DY=2" name2="..."     (container; objects of the other files come from csrc/_obj)
#   scripts/variants.sh run [bench args]                                      (GPU box; prints kernel_ms_per_step + ms_per_step per variant)
set -e
cd "$(dirname "$0")/.."
D=nabladft_amd/_variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=on -Wno-unused-function"
if [ "$1" = build ]; then
  shift; SRC=$1; shift
  rm -rf $D; mkdir -p $D
  python -m nabladft_amd.build > /dev/null
  base=$(basename $SRC .hip)
  for spec in "$@"; do
    name=${spec%%=*}; defs=${spec#*=}
    /opt/rocm/bin/hipcc $FLAGS $defs -c nabladft_amd/csrc/$SRC -o $D/${base}_$name.o &
  done
  wait
  for spec in "$@"; do
    name=${spec%%=*}
    objs=$(ls nabladft_amd/csrc/_obj/*.o | grep -v "/${base}.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libnablaq_$name.so $objs $D/${base}_$name.o
    echo "$name: ${spec#*=}" >> $D/variants.txt
  done
  rm -f $D/*.o; ls -la $D
else
  shift || true
  mkdir -p gpurun_out
  for lib in nabladft_amd/libnablaq.so $D/libnablaq_*.so; do
    echo "== $lib"
    NABLAQ_LIB=$PWD/$lib timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernel_ms_per_step'], d['ms_per_step'])"
  done 2>&1 | tee gpurun_out/variants.log
fi
