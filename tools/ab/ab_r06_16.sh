timeout 2400 python -m pytest tests/test_gpu_hp.py tests/test_gpu_hp_forced.py tests/test_gpu_fuzz.py tests/test_gpu_workspace_guard.py tests/test_gpu_fullshape_oracle.py tests/test_gpu_ops.py tests/test_gpu_extremes.py -m gpu -x -q 2>&1 | tail -5
python tools/realistic_sweep.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_realistic.txt; grep float16 gpurun_out/r06_realistic.txt
