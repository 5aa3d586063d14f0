#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
python3 -m pytest tests/test_gpu_cqt.py -m gpu -q -x > $out/tests.log 2>&1
python3 tools/cqt_bench.py > $out/cqt_mixed.txt 2>&1
BABE_FFT_MIXED=0 python3 tools/cqt_bench.py > $out/cqt_dense.txt 2>&1
tail -20 $out/tests.log; grep -v amdgpu.ids $out/cqt_mixed.txt; echo ---- dense; grep "Python loop\|whole" $out/cqt_dense.txt
