mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_modules.py -m gpu -q -x -k graph 2>&1 | tail -5 > gpurun_out/ab.txt
timeout 600 python tools/split_bench.py >> gpurun_out/ab.txt 2>&1
cat gpurun_out/ab.txt
