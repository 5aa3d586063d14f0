out=gpurun_out/r3i; mkdir -p $out
python3 -m pytest tests/test_gpu_ops.py -m gpu -q -k "nested or wino" > $out/t.log 2>&1; tail -3 $out/t.log
SHAPES=enc1.H,enc2.H0,enc2.H3,dec2.H0,dec3.H0,dec3.H4,enc3.H0,enc5.H0 python3 tools/conv_shapes_bench.py > $out/shapes_n.txt 2>&1
SHAPES=enc1.H,enc2.H0,enc2.H3,dec2.H0,dec3.H0,dec3.H4,enc3.H0,enc5.H0 BABE_CONV_WINO45=0 python3 tools/conv_shapes_bench.py > $out/shapes_4p.txt 2>&1
paste $out/shapes_n.txt $out/shapes_4p.txt | cut -c1-95,170-200
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > $out/bench.json 2> $out/bench.err; python3 -c "
import json
d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
