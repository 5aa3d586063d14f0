#!/bin/bash
out=gpurun_out/$1; shift; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_cqt.py -m gpu -q -x > $out/tests.log 2>&1; tail -3 $out/tests.log
echo "== product (analytic window)" > $out/cqt_abl.txt
BS=2,8,32 python3 tools/cqt_bench.py 2>&1 | grep "GPU time\|whole" >> $out/cqt_abl.txt
echo "== table window" >> $out/cqt_abl.txt
BABE_CQT_ANALYTIC_WIN=0 BS=32 python3 tools/cqt_bench.py 2>&1 | grep "GPU time\|whole" >> $out/cqt_abl.txt
for n in "$@"; do
  echo "== ABL=$n" >> $out/cqt_abl.txt
  BABE_HIP_LIB=$PWD/tools/abl_out/abl$n/libbabe_hip.so BS=32 python3 tools/cqt_bench.py 2>&1 | grep "GPU time" >> $out/cqt_abl.txt
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/cqt_bench.py > /dev/null 2>&1
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/cqt_kernel_stats.csv \;
rm -rf $out/trace
cat $out/cqt_abl.txt; head -8 $out/cqt_kernel_stats.csv
