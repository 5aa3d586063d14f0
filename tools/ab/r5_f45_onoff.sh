#!/bin/bash
# same-box A/B of the round's conv kernels: BABE_CONV_F45=0 = every nested layer on the round-4 F(2,5) x F(4,3) kernels
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do echo "== BABE_CONV_F45=$v"; BABE_CONV_F45=$v timeout 600 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 2>/dev/null | head -c 200 | grep -o '"value": [0-9.]*'; done
echo "== round-4 configuration: BABE_CONV_F45=0 BABE_CONV11_NT=0 BABE_CONV11P_NPW=a"
BABE_CONV_F45=0 BABE_CONV11_NT=0 BABE_CONV11P_NPW=a timeout 600 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 2>/dev/null | head -c 200 | grep -o '"value": [0-9.]*'
