timeout 2400 python -m pytest tests/test_gpu_hp.py tests/test_gpu_hp_forced.py tests/test_gpu_workspace_guard.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -4
MDCONV_HP_BWD=4 timeout 300 python tools/fuzz_more.py --seconds 200 --first 200000 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-500
