python tools/exp.py cfg3 cfg5 --label e16 --steps 20 2>&1 | grep -v amdgpu.ids
python tools/exp.py cfg3 --label e16 --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_HP_C2I=1 python tools/exp.py cfg3 --label e16-onepass --steps 20 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_gpu_hp.py tests/test_gpu_hp_forced.py tests/test_gpu_fullshape_oracle.py tests/test_analytic_pins.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -3
MDCONV_HP_C2I=1 python -m pytest tests/test_gpu_hp.py -m gpu -q -x 2>&1 | tail -2
