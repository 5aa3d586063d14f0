#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for m in 64 128 256 512; do echo "== BABE_GN_SPLITS_MAX=$m"; BABE_GN_SPLITS_MAX=$m timeout 200 python3 tools/gn_partial_bench.py 2>&1 | grep gn_partial; done
for m in 64 256 64 256; do echo "== bench BABE_GN_SPLITS_MAX=$m"; BABE_GN_SPLITS_MAX=$m timeout 600 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 2>/dev/null | head -c 200 | grep -o '"value": [0-9.]*'; done
