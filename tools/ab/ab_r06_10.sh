timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "wide_16bit_misses" 2>&1 | tail -25
