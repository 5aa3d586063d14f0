#!/bin/bash
out=gpurun_out/r5e; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_cqt.py tests/test_gpu_unet_full.py -q -x > $out/cqt_tests.log 2>&1; echo "rc=$?" >> $out/cqt_tests.log
timeout 600 python3 tools/cqt_bench.py > $out/cqt_bench.txt 2>&1
echo "---- BABE_CQT_CLIPS_PER_WG=1 (one clip per workgroup, as rounds 2-4)" >> $out/cqt_bench.txt
BS=32,64 BABE_CQT_CLIPS_PER_WG=1 timeout 600 python3 tools/cqt_bench.py >> $out/cqt_bench.txt 2>&1
echo "---- BABE_CQT_CLIPS_PER_WG=2" >> $out/cqt_bench.txt
BS=32,64 BABE_CQT_CLIPS_PER_WG=2 timeout 600 python3 tools/cqt_bench.py >> $out/cqt_bench.txt 2>&1
echo "---- BABE_CQT_CLIPS_PER_WG=8" >> $out/cqt_bench.txt
BS=32,64 BABE_CQT_CLIPS_PER_WG=8 timeout 600 python3 tools/cqt_bench.py >> $out/cqt_bench.txt 2>&1
tail -3 $out/cqt_tests.log; grep "GPU time\|whole\|----" $out/cqt_bench.txt
