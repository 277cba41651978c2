# r05 call 6: column rows out of GEMM-1's 3-D drain + dense GEMM-2 (MDCONV_BW_COLS=0: GEMM-2 re-gathers)
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cl_forced.py tests/test_analytic_pins.py tests/test_known_answers.py -m gpu -q -x 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_fullshape_oracle.py tests/test_gpu_fullsize.py -m gpu -q -x -k "cfg4 or cfg2" 2>&1 | tail -4
for i in 1 2; do
python tools/exp.py cfg4 cfg2 --label cols --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_BW_COLS=0 python tools/exp.py cfg4 cfg2 --label gather --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_BWD_FORK=0 python tools/exp.py cfg4 --label cols-nofork --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_BWD_FORK=0 MDCONV_BW_COLS=0 python tools/exp.py cfg4 --label gather-nofork --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_LIB=$PWD/modulated_deform_conv_amd/libmdconv_hip_b3old.so python tools/exp.py cfg4 cfg2 --label before --steps 20 2>&1 | grep -v amdgpu.ids
done
MDCONV_DEBUG_PLAN=1 python tools/exp.py cfg4 --label plan --steps 2 2>&1 | grep "plan:" | sort | uniq -c
} > gpurun_out/ab_r05_6.txt 2>&1
cat gpurun_out/ab_r05_6.txt
