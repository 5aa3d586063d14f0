out=gpurun_out/r3n; mkdir -p $out
timeout 600 python3 -m pytest tests/test_gpu_ops.py -m gpu -q -k "nested or wino" > $out/t.log 2>&1; tail -2 $out/t.log
SHAPES=enc3.H0,enc3.H4,enc4.H0,enc5.H0,enc5.H6,enc6.H0 timeout 300 python3 tools/conv_shapes_bench.py > $out/shapes_w2.txt 2>&1; cat $out/shapes_w2.txt
