#!/bin/bash
# usage: tools/ab/r6_fusefwd.sh <tag>: epilogue-reduction tests, engine tests, then bench ABAB of BABE_FUSE_GN_FWD=1/0
out=gpurun_out/$1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "groupnorm or f45 or wino85 or conv11 or conv1x1" > $out/t1.log 2>&1; tail -3 $out/t1.log
timeout 1500 python3 -m pytest tests/test_gpu_unet_full.py tests/test_gpu_unet_c.py -m gpu -q -x > $out/t2.log 2>&1; tail -3 $out/t2.log
for v in 1 0 1 0; do BABE_FUSE_GN_FWD=$v python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 tools/ab/jline.py FUSE_GN_FWD=$v; done | tee $out/ab.txt
