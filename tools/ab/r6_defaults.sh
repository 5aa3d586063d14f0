#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -m gpu -q -x > $out/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $out/gpu_tests.log; tail -4 $out/gpu_tests.log
BABE_EVAL_C=0 BABE_CQT_C=0 timeout 1500 python3 -m pytest tests/test_gpu_sampler.py tests/test_gpu_cqt.py tests/test_gpu_eval_c.py tests/test_gpu_flows.py -m gpu -q > $out/gpu_tests_python_sequencer.log 2>&1; echo "pytest rc=$?" >> $out/gpu_tests_python_sequencer.log; tail -4 $out/gpu_tests_python_sequencer.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
for v in "" "--no-eval-c" "" "--no-eval-c"; do python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline $v 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench $v', d['value'], d['config']['sequencer'][:40])"; done | tee $out/bench_ab.txt
