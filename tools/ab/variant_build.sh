#!/bin/bash
# usage: tools/ab/variant_build.sh <name> <source stem> <extra hipcc flags...>: tools/abl_out/libbabe_<name>.so = the product library
# with csrc/<stem>.hip rebuilt with the extra flags (same common flags as the product build); select it with BABE_HIP_LIB
set -e
cd "$(dirname "$0")/../.."
name=$1; stem=$2; shift 2
flags=$(python3 -c "import babe_amd.build as b; print(' '.join(b.COMMON_FLAGS + b.EXTRA_FLAGS.get('$stem.hip', [])))")
mkdir -p tools/abl_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags "$@" -c babe_amd/csrc/$stem.hip -o tools/abl_out/$name.$stem.o -Wno-unused-result 2>&1 | grep -v "packed-fp32-ops" || true
objs=$(ls babe_amd/build/*.hip.o | grep -v "/$stem.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-z,defs -o tools/abl_out/libbabe_$name.so $objs tools/abl_out/$name.$stem.o
ls -la tools/abl_out/libbabe_$name.so
