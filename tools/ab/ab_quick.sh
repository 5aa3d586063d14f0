#!/bin/bash
# quick A/B of ablation / variant builds on a few conv shapes: tools/ab/ab_quick.sh <out> <abl-number>...
out=gpurun_out/$1; shift; mkdir -p $out
S=${SHAPES:-enc1.H0,enc3.H0,enc4.H0,enc5.H0,enc6.H0,dec5.H0}
python3 tools/f45_check.py > $out/dbg.txt 2>&1
echo "== product" >> $out/ab.txt; SHAPES=$S python3 tools/conv_shapes_bench.py 2>&1 | grep "k=5x3\|TOTAL" >> $out/ab.txt
for n in "$@"; do
  echo "== ABL=$n" >> $out/ab.txt
  BABE_HIP_LIB=$PWD/tools/abl_out/abl$n/libbabe_hip.so python3 tools/f45_check.py 2>&1 | grep -c " OK$" >> $out/ab.txt
  SHAPES=$S BABE_HIP_LIB=$PWD/tools/abl_out/abl$n/libbabe_hip.so python3 tools/conv_shapes_bench.py 2>&1 | grep "k=5x3\|TOTAL" >> $out/ab.txt
done
echo "== product again" >> $out/ab.txt; SHAPES=$S python3 tools/conv_shapes_bench.py 2>&1 | grep "k=5x3\|TOTAL" >> $out/ab.txt
grep -v amdgpu.ids $out/dbg.txt; cat $out/ab.txt
