#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_unet_full.py -m gpu -q -s -k "full_size" > $out/full_size_tests.log 2>&1; echo "rc=$?" >> $out/full_size_tests.log
timeout 600 python3 tools/fit_kernel_ab.py > $out/fit_kernel_ab.txt 2>&1
grep -v "^$" $out/full_size_tests.log | tail -30; tail -12 $out/fit_kernel_ab.txt
