#!/bin/bash
out=gpurun_out/r5p; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python3 tools/f45_check.py > $out/f45_check.txt 2>&1
for i in 1 2; do
BABE_CONV_F45=1 timeout 600 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 > $out/bench_f45_$i.json 2> $out/bench.err
done
tail -18 $out/f45_check.txt; for f in $out/bench_*.json; do echo $f $(head -c 120 $f | grep -o '"value": [0-9.]*'); done
