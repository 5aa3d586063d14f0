#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk\|fclk" | head -5
python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --profile-steps 0 > /tmp/b.json 2>/dev/null &
BP=$!
sleep 45
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|power" | head -3; sleep 2; done
wait $BP
head -c 150 /tmp/b.json | grep -o '"value": [0-9.]*'
