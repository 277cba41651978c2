timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dgplan_forced.py tests/test_gpu_workspace_guard.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -4
timeout 200 python tools/fuzz_more.py --seconds 150 --first 220000 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-500
timeout 200 python tools/fuzz_more.py --seconds 100 --first 221000 --wide 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-500
MDCONV_QUIET=1 python tools/prof_shape.py m2:f32:B8:C200:O256:40x40:g4:dg4 m2:f32:B8:C100:O100:40x40:g4:dg4 m2:f32:B8:C320:O16:40x40:g8:dg2 m3:f32:B2:C200:O64:8x20x20:g4:dg4 m2:f16:B8:C200:O256:40x40:g4:dg4 --n 20 2>&1 | grep " ms "
