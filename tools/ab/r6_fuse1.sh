#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "partial_sums or f45 or wino85" > $out/t1.log 2>&1; tail -5 $out/t1.log
timeout 1200 python3 -m pytest tests/test_gpu_unet_full.py tests/test_gpu_unet_c.py -m gpu -q -x > $out/t2.log 2>&1; tail -5 $out/t2.log
for v in 1 0 1 0; do BABE_FUSE_GN=$v python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FUSE_GN=$v', d['value'], d['output_finite'])"; done | tee $out/ab.txt
