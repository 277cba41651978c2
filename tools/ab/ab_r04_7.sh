python -m pytest tests/test_analytic_pins.py tests/test_known_answers.py tests/test_gpu_parity.py tests/test_gpu_fullshape_oracle.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -5
MDCONV_DEBUG_PLAN=1 python tools/exp.py cfg2 --label mode2 2>&1 | grep -v amdgpu.ids | sort -u
MDCONV_BD_SPLIT=0 python tools/exp.py cfg2 --label fused 2>&1 | grep -v amdgpu.ids
MDCONV_BW_SPLITS=21 python tools/exp.py cfg2 --label mode2-s21 2>&1 | grep -v amdgpu.ids
MDCONV_BW_SPLITS=18 python tools/exp.py cfg2 --label mode2-s18 2>&1 | grep -v amdgpu.ids
MDCONV_BW_SPLITS=12 python tools/exp.py cfg2 --label mode2-s12 2>&1 | grep -v amdgpu.ids
MDCONV_BWD_FORK=0 python tools/exp.py cfg2 --label mode2-nofork 2>&1 | grep -v amdgpu.ids
MDCONV_BWD_FORK=2 python tools/exp.py cfg2 --label mode2-g2first 2>&1 | grep -v amdgpu.ids
python tools/exp.py cfg2:16 cfg2:8 cfg2:4 --label mode2 2>&1 | grep -v amdgpu.ids
MDCONV_BD_SPLIT=0 python tools/exp.py cfg2:16 cfg2:8 --label fused 2>&1 | grep -v amdgpu.ids
