# group-padded deformable groups (24 / 48 / 80 channels as 32 / 64 / 128) on the native 16-bit kernels
timeout 1500 python -m pytest tests/test_gpu_hp.py tests/test_gpu_hp_forced.py tests/test_gpu_workspace_guard.py -m gpu -x -q 2>&1 | tail -5
python tools/prof_shape.py m2:f16:B8:C96:O96:40x40:dg4 m2:f16:B8:C96:O96:40x40:dg1 m2:f16:B8:C192:O192:20x20:dg4 m2:f16:B8:C192:O192:20x20:dg1 m2:f16:B8:C320:O320:10x10:dg4 m2:bf16:B8:C96:O96:40x40:dg4 m2:f16:B16:C48:O48:56x56:dg2 m2:f16:B16:C48:O48:56x56:dg1 2>&1 | grep -v amdgpu.ids
timeout 200 python tools/fuzz_more.py --seconds 120 --first 80000 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-600
