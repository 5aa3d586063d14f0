#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in hl0 hl1; do echo "== $v"; BABE_HIP_LIB=$GRAFT_REPO_ROOT/tools/abl_out/libbabe_abl_$v.so timeout 300 python3 tools/f45_check.py 2>&1 | grep -E "C=128|C=256|bad|BAD"; done
for v in hl0 hl1 hl0 hl1; do echo "== bench $v"; BABE_HIP_LIB=$GRAFT_REPO_ROOT/tools/abl_out/libbabe_abl_$v.so timeout 600 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 2>/dev/null | head -c 200 | grep -o '"value": [0-9.]*'; done
