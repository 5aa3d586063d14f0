#!/bin/bash
# usage: abl_run.sh <shapes> <mask>...   per-shape timings of tools/conv_shapes_bench.py with the ablation builds of abl_build.sh
shapes=$1; shift
for n in "$@"; do
  echo "== ABL=$n"
  SHAPES=$shapes BABE_HIP_LIB=$PWD/tools/abl_out/abl$n/libbabe_hip.so python tools/conv_shapes_bench.py 2>&1 | grep "k=5x3"
done
