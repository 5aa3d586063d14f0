out=gpurun_out/r3r; mkdir -p $out
timeout 600 python3 -m pytest tests/test_gpu_ops.py -m gpu -q -k "nested or wino" -s > $out/t.log 2>&1; grep "nested Winograd" $out/t.log | cut -c1-150; tail -2 $out/t.log
SHAPES=enc1.H,enc2.H0,enc2.H3,dec2.H0,dec3.H0,dec3.H4 timeout 300 python3 tools/conv_shapes_bench.py > $out/shapes_w.txt 2>&1
SHAPES=enc1.H,enc2.H0,enc2.H3,dec2.H0,dec3.H0,dec3.H4 BABE_CONV_WINO45W=0 timeout 300 python3 tools/conv_shapes_bench.py > $out/shapes_n.txt 2>&1
paste $out/shapes_w.txt $out/shapes_n.txt | cut -c1-95,170-200
