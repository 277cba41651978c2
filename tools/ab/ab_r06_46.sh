# tiny-channel padding: parity (all paths), forced children, random shapes default and with the plan forced on; final-rule timings
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dgplan_forced.py tests/test_gpu_workspace_guard.py tests/test_gpu_fuzz.py tests/test_gpu_extremes.py tests/test_gpu_modules.py -m gpu -x -q 2>&1 | tail -4
MDCONV_PAD_CHANNELS=1 timeout 300 python tools/fuzz_more.py --seconds 200 --first 170000 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-500
timeout 200 python tools/fuzz_more.py --seconds 120 --first 171000 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-500
MDCONV_QUIET=1 python tools/prof_shape.py m3:f32:B2:C8:O8:16x32x32 m3:f32:B2:C64:O8:8x28x28 m2:f32:B16:C64:O8:56x56 m2:f32:B8:C8:O8:112x112 m2:f32:B8:C3:O16:112x112 d2:f32:B1:C4:O4:8x8 m3:f32:B1:C4:O4:4x8x8 m2:f32:B4:C24:O8:28x28 --n 20 2>&1 | grep " ms "
