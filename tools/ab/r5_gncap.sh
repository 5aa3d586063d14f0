#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in base cap base cap; do echo "== $v"; if [ $v = cap ]; then export BABE_HIP_LIB=$GRAFT_REPO_ROOT/tools/abl_out/libbabe_abl_gncap.so; else unset BABE_HIP_LIB; fi; timeout 600 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 2>/dev/null | head -c 200 | grep -o '"value": [0-9.]*'; done
