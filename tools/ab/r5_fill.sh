#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for f in 0.85 0.80 0.70 0.85 0.80 0.70; do echo "== BABE_W85_FILL=$f"; BABE_W85_FILL=$f timeout 600 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 2>/dev/null | head -c 200 | grep -o '"value": [0-9.]*'; done
