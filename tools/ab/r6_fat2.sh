#!/bin/bash
out=gpurun_out/$1; shift; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
echo "== product build FAT=1" > $out/fat.txt
timeout 600 python3 tools/f45_check.py 2>&1 | grep "enc3\|enc5 \|enc4\|bad" >> $out/fat.txt
for n in "$@"; do
  echo "== abl$n" >> $out/fat.txt
  BABE_HIP_LIB=$PWD/tools/abl_out/abl$n/libbabe_hip.so timeout 600 python3 tools/f45_check.py 2>&1 | grep "enc3\|enc5 \|enc4\|bad" >> $out/fat.txt
done
cat $out/fat.txt
