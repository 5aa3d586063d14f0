#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
BABE_EVAL_C=1 BABE_CQT_C=1 timeout 900 python3 -m pytest tests/test_gpu_sampler.py tests/test_gpu_cqt.py tests/test_gpu_eval_c.py -m gpu -q > $out/gpu_tests_eval_c.log 2>&1; echo "pytest rc=$?" >> $out/gpu_tests_eval_c.log
tail -4 $out/gpu_tests_eval_c.log
timeout 600 python3 tools/cqt_bench.py > $out/cqt_bench.txt 2>&1; grep "^#\|GPU time" $out/cqt_bench.txt | cut -c1-250
timeout 1500 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --precision bf16 --clips-per-gpu 64 --profile-steps 1 > $out/bench_cfg2_bf16_64clips_roofline.json 2> $out/bench_bf16.err
tail -c 300 $out/bench_cfg2_bf16_64clips_roofline.json
