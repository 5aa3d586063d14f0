#!/bin/bash
out=gpurun_out/r5p; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python3 tools/f45_check.py > $out/f45_check4.txt 2>&1
timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "winograd or f45 or conv2d" > $out/ops_tests.log 2>&1; tail -3 $out/ops_tests.log
timeout 900 python3 -m pytest tests/test_gpu_unet_full.py tests/test_gpu_sampler.py tests/test_gpu_unet_c.py -x -q -m gpu > $out/unet_tests.log 2>&1; tail -3 $out/unet_tests.log
for i in 1 2; do
timeout 600 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 > $out/bench_f45v3_$i.json 2> $out/bench.err
done
for f in $out/bench_f45v3_*.json; do echo $f $(head -c 120 $f | grep -o '"value": [0-9.]*'); done
tail -3 $out/f45_check4.txt
