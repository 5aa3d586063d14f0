out=gpurun_out/r3g; mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_unet_full.py tests/test_gpu_sampler.py -m gpu -q -x > $out/t2.log 2>&1; tail -5 $out/t2.log
python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --profile-steps 0 --precision bf16 > $out/bench_bf16_graph.json 2> $out/bench_bf16_graph.err
BABE_SAMPLER_GRAPHS=0 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --profile-steps 0 --precision bf16 > $out/bench_bf16_eager.json 2> $out/bench_bf16_eager.err
for f in bf16_graph bf16_eager; do python3 -c "
import json
d=json.loads(open('$out/bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['config']['hip_graphs'][:40])"; tail -2 $out/bench_$f.err; done
