#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
echo "== product" > $out/p96.txt
timeout 600 python3 tools/f45_check.py 2>&1 | grep "enc1\|enc2\|dec3" >> $out/p96.txt
echo "== probe: waves 4, 5 of the 96-channel tile do half their MFMAs (results wrong)" >> $out/p96.txt
BABE_HIP_LIB=$PWD/tools/abl_out/libbabe_abl_8192.so timeout 600 python3 tools/f45_check.py 2>&1 | grep "enc1\|enc2\|dec3" >> $out/p96.txt
cat $out/p96.txt
