#!/bin/bash
# usage: tools/ab/r6_bf16epi.sh <tag>: the buffer-addressed, prefetching epilogue of conv_bf16p against the library before it (prebf16):
# bf16 tests, then bench --precision bf16 (one clip, ABAB) and configs[2]'s 64 clips once each
out=gpurun_out/$1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests -m gpu -q -x -k "bf16 or units" > $out/t1.log 2>&1; tail -3 $out/t1.log
libof() { [ $1 = product ] && echo $GRAFT_REPO_ROOT/babe_amd/libbabe_hip.so || echo $GRAFT_REPO_ROOT/tools/abl_out/libbabe_$1.so; }
for v in product prebf16 product prebf16; do
  BABE_HIP_LIB=$(libof $v) timeout 600 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --precision bf16 2>/dev/null | python3 tools/ab/jline.py $v | cut -d'{' -f1
done | tee $out/bench_bf16.txt
for v in product prebf16; do
  BABE_HIP_LIB=$(libof $v) timeout 900 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --precision bf16 --clips-per-gpu 64 --profile-steps 0 2>/dev/null | python3 tools/ab/jline.py "$v-64clips" | cut -d'{' -f1
done | tee $out/bench_bf16_64.txt
