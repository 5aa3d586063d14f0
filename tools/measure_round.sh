#!/bin/bash
# Round measurement suite (run on the GPU box through gpurun): writes everything under gpurun_out/$1/ with prefix $2 (e.g. r05)
# usage: tools/measure_round.sh r6m r06
set -x
out=gpurun_out/$1
pre=${2:-r06}
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 -m pytest tests -m gpu -q -s > $out/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $out/gpu_tests.log
# the library-side UNet sequencer must not rot: the network / sampler parity files once more with BABE_UNET_C=1
BABE_UNET_C=1 python3 -m pytest tests/test_gpu_unet_c.py tests/test_gpu_unet_full.py tests/test_gpu_sampler.py -m gpu -q > $out/gpu_tests_unet_c.log 2>&1; echo "pytest rc=$?" >> $out/gpu_tests_unet_c.log
# ... since round 6 the library-side CQT plan + whole score evaluation are the DEFAULT: the Python sequencers must not rot either
BABE_EVAL_C=0 BABE_CQT_C=0 python3 -m pytest tests/test_gpu_sampler.py tests/test_gpu_cqt.py tests/test_gpu_eval_c.py tests/test_gpu_flows.py -m gpu -q > $out/gpu_tests_python_sequencer.log 2>&1; echo "pytest rc=$?" >> $out/gpu_tests_python_sequencer.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd.json 2> $out/bench.err
python3 tools/overlap_timeline.py > $out/overlap.txt 2> $out/overlap.err
python3 tools/conv_shapes_bench.py > $out/conv_shapes_fwd.txt 2>&1
VJP=1 python3 tools/conv_shapes_bench.py > $out/conv_shapes_vjp.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/trace_bench.json 2> $out/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --T 2 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $out/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --T 2 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $out/pmc_write.err
python3 tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write $out/$pre conv_wino85
SHAPES=enc3.H0,enc5.H0,enc6.H0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $out/pmc_sq -- python3 tools/conv_shapes_bench.py > /dev/null 2>&1
python3 tools/pmc_summary.py $out/pmc_sq conv_wino85 > $out/pmc_wino85.txt
SHAPES=enc3.H0,enc5.H0,enc6.H0 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA --output-format csv -d $out/pmc_insts -- python3 tools/conv_shapes_bench.py > /dev/null 2>&1
python3 tools/pmc_summary.py $out/pmc_insts conv_wino85 >> $out/pmc_wino85.txt
python3 tools/f45_check.py > $out/f45_check.txt 2>&1
python3 tools/cqt_bench.py > $out/cqt_bench.txt 2>&1
BS=1,2,8,32 rocprofv3 --kernel-trace --output-format csv -d $out/cqt_trace -- python3 tools/cqt_bench.py > /dev/null 2>&1
python3 tools/cqt_trace_summary.py $out/cqt_trace > $out/cqt_trace_summary.txt; rm -rf $out/cqt_trace
python3 tools/fit_kernel_ab.py > $out/fit_kernel_ab.txt 2>&1
python3 tools/filter_fit_bench.py > $out/filter_fit.txt 2>&1
python3 tools/host_enqueue_time.py > $out/host_enqueue_time.txt 2>&1
python3 tools/denoiser_bench.py > $out/denoiser_bench.txt 2>&1
python3 tools/config5_bench.py bf16 > $out/config5_bf16_30s.json 2> $out/config5.err
# bf16 build: one stream (the default) and the opt-in two lanes; configs[2]'s 64 clips
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --precision bf16 > $out/bench_bf16_one_stream.json 2> $out/bench_bf16.err
BABE_BF16_LANES=1 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --precision bf16 > $out/bench_bf16_two_lanes_optin.json 2>> $out/bench_bf16.err
python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --precision bf16 --clips-per-gpu 64 --profile-steps 0 > $out/bench_cfg2_bf16_64clips.json 2>> $out/bench_bf16.err
# keep the small summaries only
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/trace $out/pmc_fetch $out/pmc_write $out/pmc_sq $out/pmc_insts
ls -la $out
tail -c 400 $out/bench_driver_cmd.json
