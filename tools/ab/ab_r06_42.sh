# channel padding (pad_channels_preferred): parity incl. forced children, fuzz with the plan forced on, final rule timings
timeout 2000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dgplan_forced.py tests/test_gpu_workspace_guard.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -4
MDCONV_PAD_CHANNELS=1 timeout 300 python tools/fuzz_more.py --seconds 200 --first 160000 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-500
python tools/prof_shape.py m3:f32:B2:C32:O32:8x28x28 m3:f32:B2:C16:O16:16x32x32 m3:f32:B2:C160:O160:4x14x14 m2:f32:B16:C48:O48:56x56 m2:f32:B8:C96:O96:40x40 m3:f16:B2:C32:O32:8x28x28 --n 20 2>&1 | grep -v amdgpu.ids
