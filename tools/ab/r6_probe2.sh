#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
BABE_HIP_LIB=$GRAFT_REPO_ROOT/tools/abl_out/libbabe_probe.so timeout 300 python3 tools/f45_barrier_probe.py 2>&1 | grep -v amdgpu.ids | tee $out/barrier_probe.txt
