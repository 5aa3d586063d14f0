#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
echo "# tools/f45_ablate.py on the 12-wave form of the 128-channel tile (conv_wino85s_kernel<., 8, 12>), round 6" > $out/abl12.txt
timeout 900 python3 tools/f45_ablate.py 0 1 2 3 4 256 8 16 7 15 >> $out/abl12.txt 2>&1
echo "# the 8-wave form (BABE_W85_12W=0), same builds" >> $out/abl12.txt
BABE_W85_12W=0 timeout 900 python3 tools/f45_ablate.py 0 4 256 15 >> $out/abl12.txt 2>&1
cat $out/abl12.txt
