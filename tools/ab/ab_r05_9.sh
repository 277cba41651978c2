# r05 call 9: MFMA partial sums with the rows one step ahead (A/B: libmdconv_hip_sumsnp.so = call 8's kernel); hp_gemm2 in one round
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_gpu_hp.py tests/test_analytic_pins.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_hp_forced.py tests/test_gpu_fullshape_oracle.py -m gpu -q -x -k "cfg3 or cfg5 or chunk" 2>&1 | tail -4
NP=$PWD/modulated_deform_conv_amd/libmdconv_hip_sumsnp.so
for i in 1 2; do
python tools/exp.py cfg3 cfg5 --label pipelined --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_LIB=$NP python tools/exp.py cfg3 cfg5 --label call8-kernel --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_BWD_FORK=0 python tools/exp.py cfg3 cfg5 --label pipelined-nofork --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_BWD_FORK=0 MDCONV_LIB=$NP python tools/exp.py cfg3 cfg5 --label call8-nofork --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_HP_G2_SLOTS=3 python tools/exp.py cfg5 --label pipelined-g2-one-round --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_HP_G2_SLOTS=2 python tools/exp.py cfg5 --label pipelined-g2-two-per-cu --steps 20 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/ab_r05_9.txt 2>&1
cat gpurun_out/ab_r05_9.txt
