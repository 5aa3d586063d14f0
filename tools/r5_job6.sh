#!/bin/bash
out=gpurun_out/r5f; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python3 tools/cqt_bench.py > $out/cqt_bench.txt 2>&1
for p in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_WAVES" "TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum"; do
  n=$(echo $p | cut -c1-12 | tr ' ' '_')
  BS=32 timeout 600 rocprofv3 --kernel-trace --pmc $p --output-format csv -d $out/pmc_$n -- python3 tools/cqt_bench.py > /dev/null 2> $out/pmc_$n.err
  python3 tools/pmc_summary.py $out/pmc_$n band_fft >> $out/pmc_cqt.txt 2>&1
  python3 tools/pmc_summary.py $out/pmc_$n colfft >> $out/pmc_cqt.txt 2>&1
  rm -rf $out/pmc_$n
done
grep "GPU time\|whole" $out/cqt_bench.txt | head -12; cat $out/pmc_cqt.txt
