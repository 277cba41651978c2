python tools/exp.py cfg3 cfg5 --label fused-tabs --steps 20 2>&1 | grep -v amdgpu.ids
python tools/exp.py cfg3 --label fused-tabs --steps 20 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_gpu_hp.py tests/test_gpu_hp_forced.py tests/test_gpu_fullshape_oracle.py -m gpu -q -x 2>&1 | tail -3
