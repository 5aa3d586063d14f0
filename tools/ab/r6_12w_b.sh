#!/bin/bash
out=gpurun_out/$1; shift; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_ops.py -m gpu -q -k "f45 or wino85 or nested" > $out/ops_tests.log 2>&1; tail -3 $out/ops_tests.log
echo "== bench: product(12W) / abl builds" > $out/ab.txt
for rep in 1 2; do
  python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('product', d['value'], d['roofline']['achieved'], d['roofline']['frac'])" >> $out/ab.txt
  for n in "$@"; do
    BABE_HIP_LIB=$PWD/tools/abl_out/abl$n/libbabe_hip.so python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('abl$n', d['value'], d['roofline']['achieved'], d['roofline']['frac'])" >> $out/ab.txt
  done
done
cat $out/ab.txt
