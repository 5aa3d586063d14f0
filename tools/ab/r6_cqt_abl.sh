#!/bin/bash
# band-analysis ablations (tools/ab/abl_build.sh cqt <bits>): what do 16-byte accesses / the window loads / the FFT itself cost at B = 32?
out=gpurun_out/$1; shift; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
echo "== product" > $out/cqt_abl.txt
BS=8,32 python3 tools/cqt_bench.py 2>&1 | grep "GPU time" >> $out/cqt_abl.txt
for n in "$@"; do
  echo "== ABL=$n" >> $out/cqt_abl.txt
  BABE_HIP_LIB=$PWD/tools/abl_out/abl$n/libbabe_hip.so BS=8,32 python3 tools/cqt_bench.py 2>&1 | grep "GPU time" >> $out/cqt_abl.txt
done
cat $out/cqt_abl.txt
