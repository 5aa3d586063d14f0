out=gpurun_out/r3r; mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_unet_full.py tests/test_gpu_sampler.py -m gpu -q -x > $out/t2.log 2>&1; tail -3 $out/t2.log
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > $out/bench_w.json 2> $out/bench_w.err
BABE_CONV_WINO45W=0 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > $out/bench_n.json 2> $out/bench_n.err
for f in w n; do python3 -c "
import json
d=json.loads(open('$out/bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
