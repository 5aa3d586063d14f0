#!/bin/bash
# PMC counters of the CQT band kernels at B = 32 (three separate --pmc passes, --kernel-trace only)
out=gpurun_out/$1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
BS=32 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $out/p1 -- python3 tools/cqt_bench.py > /dev/null 2>&1
BS=32 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $out/p2 -- python3 tools/cqt_bench.py > /dev/null 2>&1
BS=32 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $out/p3 -- python3 tools/cqt_bench.py > /dev/null 2>&1
for p in p1 p2 p3; do python3 tools/pmc_summary.py $out/$p band_fft_kernel; done > $out/pmc_cqt_raw.txt
rm -rf $out/p1 $out/p2 $out/p3
cat $out/pmc_cqt_raw.txt
