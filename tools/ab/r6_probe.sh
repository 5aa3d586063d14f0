#!/bin/bash
# usage: tools/ab/r6_probe.sh <tag>: barrier-wait probe of the F(4,5) kernel, then the transform-wave priority variants (stand-alone and bench)
out=gpurun_out/$1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
BABE_HIP_LIB=$GRAFT_REPO_ROOT/tools/abl_out/libbabe_probe.so timeout 300 python3 tools/f45_barrier_probe.py 2>&1 | grep -v amdgpu.ids | tee $out/barrier_probe.txt
for v in product tprio1 tprio3; do
  lib=$GRAFT_REPO_ROOT/tools/abl_out/libbabe_$v.so; [ $v = product ] && lib=$GRAFT_REPO_ROOT/babe_amd/libbabe_hip.so
  echo "$v $(BABE_HIP_LIB=$lib timeout 300 python3 tools/f45_ablate.py child 2>/dev/null)"
done | tee $out/tprio_standalone.txt
for v in product tprio3 product tprio3; do
  lib=$GRAFT_REPO_ROOT/tools/abl_out/libbabe_$v.so; [ $v = product ] && lib=$GRAFT_REPO_ROOT/babe_amd/libbabe_hip.so
  BABE_HIP_LIB=$lib python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 tools/ab/jline.py $v
done | tee $out/tprio_bench.txt
