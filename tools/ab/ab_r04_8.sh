python -m pytest tests/test_gpu_parity.py tests/test_analytic_pins.py tests/test_gpu_hp_forced.py tests/test_gpu_ops.py tests/test_gpu_fullsize.py -m gpu -q 2>&1 | tail -12
python tools/exp.py cfg2 cfg4 --label now 2>&1 | grep -v amdgpu.ids
python tools/exp.py cfg2 --label now2 2>&1 | grep -v amdgpu.ids
