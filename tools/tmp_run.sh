out=gpurun_out/r3h; mkdir -p $out
python3 -m pytest tests/test_gpu_ops.py -m gpu -q -k "nested or wino or resample" > $out/t.log 2>&1; tail -3 $out/t.log
SHAPES=enc0.H0,enc3.H0,enc4.H0,enc5.H0,enc6.H0 python3 tools/conv_shapes_bench.py > $out/shapes.txt 2>&1; cat $out/shapes.txt
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > $out/bench.json 2> $out/bench.err; python3 -c "
import json
d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
PRECISION=f32 python3 tools/host_enqueue_time.py > $out/host_enqueue_time.txt 2>&1; PRECISION=bf16 python3 tools/host_enqueue_time.py >> $out/host_enqueue_time.txt 2>&1; cat $out/host_enqueue_time.txt
