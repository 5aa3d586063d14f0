#!/bin/bash
# usage: tools/ab/r6_head.sh <tag>: bench A/B of the working tree (BABE_FUSE_GN=0 and =1) against the HEAD copy in tools/abl_out/head
out=gpurun_out/$1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { ( cd $2; BABE_FUSE_GN=$3 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>$GRAFT_REPO_ROOT/$out/err_$1.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['output_finite'])" ); }
for i in 1 2; do
  run head tools/abl_out/head 0; run fuse0 . 0; run fuse1 . 1
done | tee $out/ab.txt
tail -3 $out/err_head.log
