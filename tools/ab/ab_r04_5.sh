python -m pytest tests/test_analytic_pins.py tests/test_known_answers.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -8
python tools/exp.py cfg2 cfg4 --label split 2>&1 | grep -v amdgpu.ids
MDCONV_BWD_FORK=2 python tools/exp.py cfg2 cfg4 --label split-gemm2first 2>&1 | grep -v amdgpu.ids
MDCONV_BD_SPLIT=0 python tools/exp.py cfg2 cfg4 --label fused 2>&1 | grep -v amdgpu.ids
MDCONV_BWD_FORK=0 python tools/exp.py cfg2 cfg4 --label split-nofork 2>&1 | grep -v amdgpu.ids
