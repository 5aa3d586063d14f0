#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
echo "== 12W=1" > $out/w12.txt
BABE_W85_12W=1 timeout 600 python3 tools/f45_check.py 2>&1 | grep -v amdgpu >> $out/w12.txt
echo "== 12W=0" >> $out/w12.txt
timeout 600 python3 tools/f45_check.py 2>&1 | grep "enc\|dec\|bad" >> $out/w12.txt
echo "== bench 12W=1 / 0 / 1 / 0" >> $out/w12.txt
for v in 1 0 1 0; do BABE_W85_12W=$v python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('12W=$v', d['value'], d['roofline']['achieved'], d['roofline']['frac'])" >> $out/w12.txt; done
cat $out/w12.txt
