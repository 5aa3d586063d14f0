#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python3 -m pytest tests/test_gpu_sampler.py -m gpu -q -s -k "sweep or helper or diagnostics or inpainting" > $out/new_tests.log 2>&1; echo "rc=$?" >> $out/new_tests.log
grep -v "^$\|amdgpu.ids" $out/new_tests.log | tail -40
