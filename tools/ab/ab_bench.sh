#!/bin/bash
# same-box A/B of whole-job throughput: product library vs variant libraries (tools/abl_out/<name>/libbabe_hip.so)
out=gpurun_out/$1; shift; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > $out/bench_product.json 2> $out/err_product.txt
for n in "$@"; do
  BABE_HIP_LIB=$PWD/tools/abl_out/$n/libbabe_hip.so python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > $out/bench_$n.json 2> $out/err_$n.txt
done
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > $out/bench_product2.json 2>> $out/err_product.txt
for f in $out/bench_*.json; do echo $f $(python3 -c "import json,sys; d=json.loads(open('$f').read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'])"); done
