#!/bin/bash
out=gpurun_out/r5h; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_cqt.py -q -x -s > $out/cqt_tests.log 2>&1; echo "rc=$?" >> $out/cqt_tests.log
BS=1,2,8,32,64 timeout 600 python3 tools/cqt_bench.py > $out/cqt_bench.txt 2>&1
echo "---- BABE_FFT_REAL=0 (complex form on both stages, as round 4 but 7-column tiles)" >> $out/cqt_bench.txt
BS=2,32 BABE_FFT_REAL=0 timeout 600 python3 tools/cqt_bench.py >> $out/cqt_bench.txt 2>&1
timeout 2400 python3 -m pytest tests -m gpu -q > $out/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $out/gpu_tests.log
tail -5 $out/cqt_tests.log; grep "GPU time\|whole\|Python\|----" $out/cqt_bench.txt; tail -4 $out/gpu_tests.log
