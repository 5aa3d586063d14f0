#!/bin/bash
# usage: tools/ab/r6_fusefwd_prof.sh <tag>: kernel-trace stats of one bench step with BABE_FUSE_GN_FWD=0 (A) and =1 (B), side by side
out=gpurun_out/$1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export BABE_FUSE_GN_FWD=0
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/pA -o a -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/a.log 2>&1
export BABE_FUSE_GN_FWD=1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/pB -o b -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/b.log 2>&1
python3 tools/kstats_diff.py $out/pA $out/pB 30 | tee $out/diff.txt
find $out -name "*kernel_trace.csv" -delete; find $out -name "*.db" -delete
