#!/bin/bash
# A/B of the second-generation wide nested-Winograd kernel (conv_wino45x) against the first (BABE_CONV_WINO45X=0), same box.
# usage (through gpurun): tools/ab/ab_wino45x.sh <outdir-name>
out=gpurun_out/$1
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "nested or wino" > $out/tests.log 2>&1; echo "pytest rc=$?" >> $out/tests.log
python3 tools/conv_shapes_bench.py > $out/shapes_x1.txt 2>&1
BABE_CONV_WINO45X=0 python3 tools/conv_shapes_bench.py > $out/shapes_x0.txt 2>&1
VJP=1 python3 tools/conv_shapes_bench.py > $out/shapes_vjp_x1.txt 2>&1
VJP=1 BABE_CONV_WINO45X=0 python3 tools/conv_shapes_bench.py > $out/shapes_vjp_x0.txt 2>&1
SHAPES=enc3.H0,enc5.H0,enc6.H0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $out/pmc_sq -- python3 tools/conv_shapes_bench.py > /dev/null 2>&1
python3 tools/pmc_summary.py $out/pmc_sq conv_wino45 > $out/pmc_wino45x.txt
rm -rf $out/pmc_sq
tail -3 $out/tests.log; tail -1 $out/shapes_x1.txt; tail -1 $out/shapes_x0.txt; tail -1 $out/shapes_vjp_x1.txt; tail -1 $out/shapes_vjp_x0.txt; cat $out/pmc_wino45x.txt
