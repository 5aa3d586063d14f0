#!/bin/bash
# same-box A/B/A/B of whole-job throughput under an environment switch: tools/ab/ab_env_bench.sh <out> "<ENV=VAL>"
out=gpurun_out/$1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { env "$@" python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'])"; }
echo "default  $(run X=1)" | tee -a $out/ab.txt
echo "$2  $(run $2)" | tee -a $out/ab.txt
echo "default  $(run X=1)" | tee -a $out/ab.txt
echo "$2  $(run $2)" | tee -a $out/ab.txt
