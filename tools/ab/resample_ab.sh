#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
python3 -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "resample or block" > $out/tests.log 2>&1
python3 tools/resample_bench.py > $out/resample.txt 2>&1
tail -3 $out/tests.log; grep -v amdgpu $out/resample.txt
