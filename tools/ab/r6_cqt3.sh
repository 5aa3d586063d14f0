#!/bin/bash
out=gpurun_out/$1; shift; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
echo "== product" > $out/cqt_abl.txt
BS=32 python3 tools/cqt_bench.py 2>&1 | grep "GPU time\|whole" >> $out/cqt_abl.txt
for n in "$@"; do
  echo "== ABL=$n" >> $out/cqt_abl.txt
  BABE_HIP_LIB=$PWD/tools/abl_out/abl$n/libbabe_hip.so BS=32 python3 tools/cqt_bench.py 2>&1 | grep "GPU time" >> $out/cqt_abl.txt
done
BS=1,2,8,32 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 tools/cqt_bench.py > /dev/null 2>&1
python3 tools/cqt_trace_summary.py $out/trace > $out/cqt_trace_summary.txt
head -3 $(find $out/trace -name "*kernel_trace.csv" | head -1) > $out/trace_head.txt
rm -rf $out/trace
cat $out/cqt_abl.txt $out/cqt_trace_summary.txt; cat $out/trace_head.txt | cut -c1-600
