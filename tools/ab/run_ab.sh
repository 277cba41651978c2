mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_hp.py tests/test_gpu_modules.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/ab.txt
timeout 600 python tools/split_bench.py >> gpurun_out/ab.txt 2>&1
cat gpurun_out/ab.txt
