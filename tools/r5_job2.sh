#!/bin/bash
out=gpurun_out/r5b; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python3 tools/overlap_timeline.py > $out/overlap.txt 2> $out/overlap.err
timeout 900 python3 -m pytest tests/test_gpu_dist.py -x -q > $out/dist_tests.log 2>&1
timeout 600 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 > $out/bench_eager.json 2> $out/bench_eager.err
BABE_SAMPLER_GRAPHS=1 timeout 900 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 > $out/bench_graphs.json 2> $out/bench_graphs.err
timeout 600 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 > $out/bench_eager2.json 2> $out/bench_eager2.err
tail -5 $out/dist_tests.log; head -c 300 $out/bench_eager.json; echo; head -c 300 $out/bench_graphs.json; echo; head -c 300 $out/bench_eager2.json
