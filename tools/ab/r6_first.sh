#!/bin/bash
# round-6 first GPU call: baseline numbers of this round's tree (bench + cqt + fit-kernel A/B)
out=gpurun_out/r6a; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 bench.py --steps 4 --warmup 2 > $out/bench.json 2> $out/bench.err
python3 tools/cqt_bench.py > $out/cqt_bench.txt 2>&1
python3 tools/fit_kernel_ab.py > $out/fit_kernel_ab.txt 2>&1
tail -c 600 $out/bench.json; cat $out/cqt_bench.txt | grep GPU; cat $out/fit_kernel_ab.txt | tail -8
