#!/bin/bash
out=gpurun_out/r5d; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests -m gpu -q -s > $out/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $out/gpu_tests.log
tail -8 $out/gpu_tests.log
