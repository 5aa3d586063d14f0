#!/bin/bash
out=gpurun_out/r5g; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_cqt.py -q -x > $out/cqt_tests.log 2>&1; echo "rc=$?" >> $out/cqt_tests.log
BS=1,2,32 timeout 600 python3 tools/cqt_bench.py > $out/cqt_bench.txt 2>&1
echo "---- BABE_FFT_REAL=0" >> $out/cqt_bench.txt
BS=1,2,32 BABE_FFT_REAL=0 timeout 600 python3 tools/cqt_bench.py >> $out/cqt_bench.txt 2>&1
tail -5 $out/cqt_tests.log; grep "GPU time\|whole\|----" $out/cqt_bench.txt
