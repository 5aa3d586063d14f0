#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in 0 1 2 3 0 3; do BABE_ABL_GN=$v python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ABL_GN=$v', d['value'], d['output_finite'])"; done | tee $out/gnprobe.txt
