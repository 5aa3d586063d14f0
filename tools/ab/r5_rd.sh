#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rd in 12 6 3 12 6 3; do echo "== ring depth $rd"; BABE_HIP_LIB=$GRAFT_REPO_ROOT/tools/abl_out/libbabe_abl_$rd.so timeout 600 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 2>/dev/null | head -c 200 | grep -o '"value": [0-9.]*'; done
