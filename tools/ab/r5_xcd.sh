#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_unet_full.py tests/test_gpu_unet_c.py -q -m gpu 2>&1 | tail -3
for x in 1 0 1 0; do echo "== BABE_W85_XCD=$x"; BABE_W85_XCD=$x timeout 600 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 2>/dev/null | head -c 200 | grep -o '"value": [0-9.]*'; done
