timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dgplan_forced.py tests/test_gpu_workspace_guard.py tests/test_gpu_fuzz.py tests/test_gpu_extremes.py -m gpu -x -q 2>&1 | tail -4
timeout 200 python tools/fuzz_more.py --seconds 150 --first 190000 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-500
MDCONV_QUIET=1 python tools/prof_shape.py m2:f32:B8:C100:O100:40x40 m2:f32:B8:C256:O4:40x40:dg2 m2:f32:B8:C16:O256:40x40:dg4 m2:f32:B8:C8:O16:40x40:dg4 m3:f32:B2:C256:O4:8x20x20:dg4 m3:f32:B2:C16:O256:8x20x20:dg4 m2:f32:B8:C36:O36:40x40 --n 20 2>&1 | grep " ms "
