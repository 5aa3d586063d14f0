#!/bin/bash
# usage: tools/ab/r6_fuse2.sh <tag>: f45 epilogue tests, then bench A/B/C: working tree with BABE_FUSE_GN=1 / =0, and the HEAD copy
# under tools/abl_out/head (built beforehand).
out=gpurun_out/$1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "partial_sums or f45 or wino85" > $out/t1.log 2>&1; tail -3 $out/t1.log
timeout 1200 python3 -m pytest tests/test_gpu_unet_full.py tests/test_gpu_unet_c.py -m gpu -q -x > $out/t2.log 2>&1; tail -3 $out/t2.log
run() { ( cd $2; BABE_FUSE_GN=$3 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['output_finite'])" ); }
for i in 1 2; do
  run fuse1 . 1; run fuse0 . 0; run head tools/abl_out/head 0
done | tee $out/ab.txt
