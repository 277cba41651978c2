# matrix-core partial sums for lists of 16 channels (half a 32-column block)
timeout 1200 python -m pytest tests/test_gpu_hp.py tests/test_gpu_hp_forced.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -4
python tools/prof_shape.py m2:f16:B16:C64:O64:56x56:dg4 m2:f16:B8:C64:O256:56x56:dg4 m2:f16:B16:C32:O32:56x56:dg2 m2:f16:B16:C64:O64:56x56:dg1 2>&1 | grep -v amdgpu.ids
MDCONV_HP_BWD=4 timeout 200 python tools/fuzz_more.py --seconds 120 --first 70000 --dg 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-400
