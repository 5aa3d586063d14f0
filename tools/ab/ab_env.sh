#!/bin/bash
# same-box A/B of conv shapes under environment switches: tools/ab/ab_env.sh <out> "<ENV=VAL ...>" ...
out=gpurun_out/$1; shift; mkdir -p $out
S=${SHAPES:-enc1.H0,enc2.H0,enc3.H0,enc4.H0,enc5.H0,enc6.H0,dec5.H0}
python3 tools/f45_check.py > $out/dbg.txt 2>&1
echo "== default" >> $out/ab.txt; SHAPES=$S python3 tools/conv_shapes_bench.py 2>&1 | grep "k=5x3\|TOTAL" >> $out/ab.txt
for e in "$@"; do
  echo "== $e" >> $out/ab.txt
  env $e SHAPES=$S python3 tools/conv_shapes_bench.py 2>&1 | grep "k=5x3\|TOTAL" >> $out/ab.txt
done
echo "== default again" >> $out/ab.txt; SHAPES=$S python3 tools/conv_shapes_bench.py 2>&1 | grep "k=5x3\|TOTAL" >> $out/ab.txt
grep -v amdgpu.ids $out/dbg.txt; cat $out/ab.txt
