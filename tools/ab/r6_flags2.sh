#!/bin/bash
# usage: tools/ab/r6_flags2.sh <tag> <variant>...: correctness gate, stand-alone launch times (two passes) and one bench run per variant, twice
out=gpurun_out/$1; mkdir -p $out; shift
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
libof() { [ $1 = product ] && echo $GRAFT_REPO_ROOT/babe_amd/libbabe_hip.so || echo $GRAFT_REPO_ROOT/tools/abl_out/libbabe_$1.so; }
ok=""
for v in "$@"; do
  if [ $v != product ]; then
    BABE_HIP_LIB=$(libof $v) timeout 240 python3 tools/f45_check.py > $out/check_$v.txt 2>&1
    n=$(grep -c " OK" $out/check_$v.txt)
    if [ "$n" != "11" ] || grep -q "BAD" $out/check_$v.txt; then echo "$v: check failed ($n OK)"; continue; fi
  fi
  ok="$ok $v"
done
for i in 1 2; do for v in $ok; do echo "$v $(BABE_HIP_LIB=$(libof $v) timeout 300 python3 tools/f45_ablate.py child 2>/dev/null)"; done; done | tee $out/standalone.txt
for i in 1 2; do for v in $ok; do BABE_HIP_LIB=$(libof $v) timeout 600 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 tools/ab/jline.py $v | cut -d'{' -f1; done; done | tee $out/bench.txt
