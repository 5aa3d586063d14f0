#!/bin/bash
out=gpurun_out/r5q; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
SHAPES=enc0.H0,enc1.H0,dec3.H0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $out/pmc_sq -- python3 tools/conv_shapes_bench.py > /dev/null 2>&1
python3 tools/pmc_summary.py $out/pmc_sq conv_wino85s > $out/pmc_wino85s.txt
SHAPES=enc0.H0,enc1.H0,dec3.H0 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA --output-format csv -d $out/pmc_insts -- python3 tools/conv_shapes_bench.py > /dev/null 2>&1
python3 tools/pmc_summary.py $out/pmc_insts conv_wino85s >> $out/pmc_wino85s.txt
rm -rf $out/pmc_sq $out/pmc_insts
cat $out/pmc_wino85s.txt
