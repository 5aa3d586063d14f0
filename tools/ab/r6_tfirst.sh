#!/bin/bash
# usage: tools/ab/r6_tfirst.sh <tag>: transform waves as the oldest (product) vs the youngest (tlast) waves of the workgroup:
# parity tests, barrier probe, stand-alone launch times, bench ABAB
out=gpurun_out/$1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "groupnorm or f45 or wino85" > $out/t1.log 2>&1; tail -2 $out/t1.log
BABE_HIP_LIB=$GRAFT_REPO_ROOT/tools/abl_out/libbabe_probe.so timeout 300 python3 tools/f45_barrier_probe.py 2>&1 | grep -v amdgpu.ids | tee $out/barrier_probe.txt
for v in product tlast; do
  lib=$GRAFT_REPO_ROOT/tools/abl_out/libbabe_$v.so; [ $v = product ] && lib=$GRAFT_REPO_ROOT/babe_amd/libbabe_hip.so
  echo "$v $(BABE_HIP_LIB=$lib timeout 300 python3 tools/f45_ablate.py child 2>/dev/null)"
done | tee $out/standalone.txt
for v in product tlast product tlast; do
  lib=$GRAFT_REPO_ROOT/tools/abl_out/libbabe_$v.so; [ $v = product ] && lib=$GRAFT_REPO_ROOT/babe_amd/libbabe_hip.so
  BABE_HIP_LIB=$lib python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 tools/ab/jline.py $v
done | tee $out/bench.txt
