#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_eval_c.py -m gpu -q -x -s > $out/eval_c_tests.log 2>&1; echo "rc=$?" >> $out/eval_c_tests.log
grep -v "^$" $out/eval_c_tests.log | tail -40
