#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for n in auto 1 2 auto 1 2; do echo "== BABE_CONV11P_NPW=$n"; if [ $n = auto ]; then unset BABE_CONV11P_NPW; else export BABE_CONV11P_NPW=$n; fi; timeout 600 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 2>/dev/null | head -c 200 | grep -o '"value": [0-9.]*'; done
