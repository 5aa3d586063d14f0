#!/bin/bash
# builds libbabe_hip.so variants with -DABL=n for conv_bf16p.hip into tools/scratch/abl<n>/babe_amd-like dirs
set -e
cd /root/repo
for n in "$@"; do
  mkdir -p tools/scratch/abl$n
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DABL=$n -c babe_amd/csrc/conv_bf16p.hip -o tools/scratch/abl$n/conv_bf16p.o -Wno-unused-result
  objs=$(ls babe_amd/build/*.hip.o | grep -v conv_bf16p)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-z,defs -o tools/scratch/abl$n/libbabe_hip.so $objs tools/scratch/abl$n/conv_bf16p.o
done
