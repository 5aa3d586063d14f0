#!/bin/bash
# usage: abl_build.sh <file-stem> <n>...   builds tools/abl_out/abl<n>/libbabe_hip.so with -DABL=n for csrc/<stem>.hip
# (same compile flags as the product build: babe_amd.build.COMMON_FLAGS - no packed-fp32 instructions)
set -e
cd "$(dirname "$0")/../.."
stem=$1; shift
flags=$(python3 -c "import babe_amd.build as b; print(' '.join(b.COMMON_FLAGS + b.EXTRA_FLAGS.get('$stem.hip', [])))")
for n in "$@"; do
  mkdir -p tools/abl_out/abl$n
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -DABL=$n $ABL_FLAGS -c babe_amd/csrc/$stem.hip -o tools/abl_out/abl$n/$stem.o -Wno-unused-result 2>&1 | grep -v "packed-fp32-ops" || true
  objs=$(ls babe_amd/build/*.hip.o | grep -v "/$stem.hip.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-z,defs -o tools/abl_out/abl$n/libbabe_hip.so $objs tools/abl_out/abl$n/$stem.o
done
