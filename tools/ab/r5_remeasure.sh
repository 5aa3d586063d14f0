#!/bin/bash
# the conv-kernel part of tools/measure_round.sh once more (after the last kernel change of the round)
out=gpurun_out/r5r; pre=r05; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 tools/conv_shapes_bench.py > $out/conv_shapes_fwd.txt 2>&1
VJP=1 python3 tools/conv_shapes_bench.py > $out/conv_shapes_vjp.txt 2>&1
python3 tools/f45_check.py > $out/f45_check.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/trace_bench.json 2> $out/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --T 2 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $out/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --T 2 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $out/pmc_write.err
python3 tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write $out/$pre conv_wino85
SHAPES=enc3.H0,enc5.H0,enc6.H0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $out/pmc_sq -- python3 tools/conv_shapes_bench.py > /dev/null 2>&1
python3 tools/pmc_summary.py $out/pmc_sq conv_wino85_kernel > $out/pmc_wino85.txt
SHAPES=enc3.H0,enc5.H0,enc6.H0 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA --output-format csv -d $out/pmc_insts -- python3 tools/conv_shapes_bench.py > /dev/null 2>&1
python3 tools/pmc_summary.py $out/pmc_insts conv_wino85_kernel >> $out/pmc_wino85.txt
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/trace $out/pmc_fetch $out/pmc_write $out/pmc_sq $out/pmc_insts
cat $out/$pre"_conv_traffic.json"; cat $out/pmc_wino85.txt; head -8 $out/kernel_stats.csv | cut -c1-160; tail -2 $out/conv_shapes_fwd.txt
