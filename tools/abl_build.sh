#!/bin/bash
# usage: abl_build.sh <file-stem> <n>...   builds tools/abl_out/abl<n>/libbabe_hip.so with -DABL=n for csrc/<stem>.hip
set -e
cd /root/repo
stem=$1; shift
for n in "$@"; do
  mkdir -p tools/abl_out/abl$n
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DABL=$n -c babe_amd/csrc/$stem.hip -o tools/abl_out/abl$n/$stem.o -Wno-unused-result
  objs=$(ls babe_amd/build/*.hip.o | grep -v "/$stem.hip.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-z,defs -o tools/abl_out/abl$n/libbabe_hip.so $objs tools/abl_out/abl$n/$stem.o
done
