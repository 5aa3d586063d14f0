out=gpurun_out/r3f; mkdir -p $out
python3 -m pytest tests/test_gpu_cqt.py -m gpu -q -x > $out/t.log 2>&1; tail -5 $out/t.log
BS=2,32,64 python3 tools/cqt_bench.py > $out/cqt_bench2.txt 2>&1; cat $out/cqt_bench2.txt
