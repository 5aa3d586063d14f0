#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for t in 1 0; do echo "== BABE_W85_TPW=$t"; BABE_W85_TPW=$t timeout 300 python3 tools/f45_check.py 2>&1 | grep -E "^F45|^enc|^dec|bad"; done
