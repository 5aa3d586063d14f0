#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
python3 tools/w45x_debug.py > $out/dbg.txt 2>&1
python3 -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "nested or wino" >> $out/dbg.txt 2>&1
for i in 1 2 3; do python3 tools/w45x_debug.py 2>&1 | grep -c "bad elements 0" >> $out/dbg.txt; done
cat $out/dbg.txt
