#!/bin/bash
# builds the kernel-lab executables (in-tree, git-ignored; they travel to the GPU box with the snapshot)
set -e
cd "$(dirname "$0")/../.."
mkdir -p scripts/lab/_bin
for f in scripts/lab/*_lab.hip; do
  n=$(basename $f .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I nabladft_amd/csrc $f -o scripts/lab/_bin/$n -L nabladft_amd -lnablaq -Wl,-rpath,'$ORIGIN/../../../nabladft_amd' &
done
wait
ls -la scripts/lab/_bin
