#!/bin/bash
out=gpurun_out/r5c; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python3 tools/overlap_timeline.py > $out/overlap.txt 2> $out/overlap.err
timeout 1500 python3 -m pytest tests -m gpu -x -q -s > $out/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $out/gpu_tests.log
timeout 600 python3 tools/config5_bench.py bf16 > $out/config5_bf16.json 2> $out/config5.err
tail -5 $out/gpu_tests.log; tail -3 $out/overlap.err; cat $out/config5_bf16.json | tail -1
