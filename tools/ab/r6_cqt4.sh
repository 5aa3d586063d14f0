#!/bin/bash
out=gpurun_out/$1; shift; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_cqt.py -m gpu -q -x > $out/tests.log 2>&1; tail -15 $out/tests.log
BS=1,2,8,32,64 python3 tools/cqt_bench.py > $out/cqt_bench.txt 2>&1
BS=1,2,8,32 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 tools/cqt_bench.py > /dev/null 2>&1
python3 tools/cqt_trace_summary.py $out/trace > $out/cqt_trace_summary.txt
rm -rf $out/trace
grep "GPU time\|whole" $out/cqt_bench.txt; cat $out/cqt_trace_summary.txt
