#!/bin/bash
out=gpurun_out/r5p; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
echo "== XCD order on" > $out/f45_xcd.txt
timeout 300 python3 tools/f45_ablate.py 0 32 >> $out/f45_xcd.txt 2>&1
echo "== XCD order off" >> $out/f45_xcd.txt
BABE_W85_XCD=0 timeout 300 python3 tools/f45_ablate.py 0 32 >> $out/f45_xcd.txt 2>&1
timeout 600 python3 tools/f45_check.py > $out/f45_check2.txt 2>&1
cat $out/f45_xcd.txt; tail -17 $out/f45_check2.txt
