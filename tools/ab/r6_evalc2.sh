#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_unet_full.py -m gpu -q -s -k "library_evaluation" > $out/full_size_evalc.log 2>&1; echo "rc=$?" >> $out/full_size_evalc.log
grep -v "^$\|amdgpu" $out/full_size_evalc.log | tail -8
for v in "" "--eval-c" "" "--eval-c"; do python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline $v 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench $v', d['value'], d['config']['sequencer'][:30])"; done | tee $out/bench_evalc.txt
