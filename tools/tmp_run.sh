out=gpurun_out/r3k; mkdir -p $out
python3 -m pytest tests/test_gpu_ops.py -m gpu -q -k "nested or wino" -s > $out/t.log 2>&1; grep "nested Winograd" $out/t.log | cut -c1-150; tail -2 $out/t.log
SHAPES=enc4.H5,enc5.H5,enc5.H6,enc6.H0,enc6.H4,enc6.H5,enc6.H6 python3 tools/conv_shapes_bench.py > $out/shapes_n.txt 2>&1; cat $out/shapes_n.txt
python3 -m pytest tests/test_gpu_unet_full.py -m gpu -q -x > $out/t2.log 2>&1; tail -2 $out/t2.log
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > $out/bench.json 2> $out/bench.err; python3 -c "
import json
d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
