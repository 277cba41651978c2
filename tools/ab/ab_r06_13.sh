timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_hp.py tests/test_gpu_hp_forced.py tests/test_gpu_fullshape_oracle.py -m gpu -x -q 2>&1 | tail -4
python tools/exp.py cfg3 cfg5 --label mb4 --steps 20 2>&1 | grep -v amdgpu.ids
python tools/prof_shape.py m2:f16:B8:C256:O256:56x56:dg1 m2:f16:B32:C256:O256:56x56:dg1 m2:f16:B2:C256:O256:56x56:dg1 m2:f16:B16:C256:O256:14x14:dg1 m2:f16:B16:C512:O512:7x7:dg1 m3:f16:B4:C256:O256:4x7x7:dg1 2>&1 | grep -v amdgpu.ids
