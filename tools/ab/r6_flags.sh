#!/bin/bash
# usage: tools/ab/r6_flags.sh <tag>: the barrier-free (flag-synchronised, three X buffers) variant of the specialised-wave F(4,5) kernels:
# correctness (tools/f45_check.py, bit-identity tests), stand-alone launch times, bench ABAB.  Every step under its own timeout.
out=gpurun_out/$1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/tools/abl_out/libbabe_flags.so
BABE_HIP_LIB=$V timeout 240 python3 tools/f45_check.py > $out/check.txt 2>&1; echo "check rc=$?"; grep -c OK $out/check.txt; grep -v OK $out/check.txt | grep -v amdgpu.ids | head -5
if ! grep -q "OK" $out/check.txt || grep -q "BAD\|Error\|error" $out/check.txt; then echo "variant is wrong or hung: stopping"; exit 0; fi
BABE_HIP_LIB=$V timeout 600 python3 -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "groupnorm or f45 or wino85" 2>&1 | tail -2
for v in product flags; do
  lib=$GRAFT_REPO_ROOT/tools/abl_out/libbabe_$v.so; [ $v = product ] && lib=$GRAFT_REPO_ROOT/babe_amd/libbabe_hip.so
  echo "$v $(BABE_HIP_LIB=$lib timeout 300 python3 tools/f45_ablate.py child 2>/dev/null)"
done | tee $out/standalone.txt
for v in product flags product flags; do
  lib=$GRAFT_REPO_ROOT/tools/abl_out/libbabe_$v.so; [ $v = product ] && lib=$GRAFT_REPO_ROOT/babe_amd/libbabe_hip.so
  BABE_HIP_LIB=$lib timeout 600 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 tools/ab/jline.py $v
done | tee $out/bench.txt
