#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
python3 tools/filter_fit_bench.py > $out/fit.txt 2>&1
BABE_FIT_FAST=0 python3 tools/filter_fit_bench.py >> $out/fit.txt 2>&1
python3 -m pytest tests/test_gpu_stft.py tests/test_gpu_sampler.py -m gpu -q -x > $out/tests.log 2>&1
cat $out/fit.txt; tail -5 $out/tests.log
