out=gpurun_out/r3j; mkdir -p $out
python3 -m pytest tests/test_gpu_ops.py -m gpu -q -k "nested or wino" -s > $out/t.log 2>&1; grep "nested Winograd" $out/t.log | cut -c1-150; tail -2 $out/t.log
SHAPES=enc4.H3,enc4.H4,enc4.H5,enc5.H4,enc5.H5,enc5.H6 python3 tools/conv_shapes_bench.py > $out/shapes_n.txt 2>&1
SHAPES=enc4.H3,enc4.H4,enc4.H5,enc5.H4,enc5.H5,enc5.H6 BABE_CONV_WINO45=0 python3 tools/conv_shapes_bench.py > $out/shapes_4p.txt 2>&1
paste $out/shapes_n.txt $out/shapes_4p.txt | cut -c1-95,170-200
