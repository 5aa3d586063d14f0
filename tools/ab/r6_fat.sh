#!/bin/bash
# fat-wave F(4,5) kernel: correctness + stand-alone timing against the 8-wave kernel (BABE_W85_FAT=0)
out=gpurun_out/$1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
echo "== FAT=1" > $out/fat.txt
timeout 600 python3 tools/f45_check.py >> $out/fat.txt 2>&1
echo "== FAT=0" >> $out/fat.txt
BABE_W85_FAT=0 timeout 600 python3 tools/f45_check.py >> $out/fat.txt 2>&1
grep -v amdgpu.ids $out/fat.txt
