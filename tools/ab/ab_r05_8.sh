# r05 call 8: per-anchor partial sums on the matrix cores (MDCONV_HP_SUMS=0: the VALU kernel)
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_gpu_hp.py tests/test_analytic_pins.py tests/test_known_answers.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_hp_forced.py tests/test_gpu_fullshape_oracle.py -m gpu -q -x -k "cfg3 or cfg5 or chunk" 2>&1 | tail -4
for i in 1 2; do
python tools/exp.py cfg3 cfg5 --label mfma-sums --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_HP_SUMS=0 python tools/exp.py cfg3 cfg5 --label valu-sums --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_BWD_FORK=0 python tools/exp.py cfg3 cfg5 --label mfma-sums-nofork --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_BWD_FORK=0 MDCONV_HP_SUMS=0 python tools/exp.py cfg3 cfg5 --label valu-sums-nofork --steps 20 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/ab_r05_8.txt 2>&1
cat gpurun_out/ab_r05_8.txt
