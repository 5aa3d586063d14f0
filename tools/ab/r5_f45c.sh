#!/bin/bash
out=gpurun_out/r5p; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python3 tools/f45_check.py > $out/f45_check3.txt 2>&1
tail -22 $out/f45_check3.txt
