# r05 call 19: phase breakdowns (s_memtime stamps) of hp_fwd2 / hp_bwd3 / GEMM-1 / GEMM-2 at cfg5 / cfg3 / cfg4; GEMM-1 ablations
# at cfg4 (no corner gathers, no row stores); hp_bwd3 row-store experiments (grad_col piece stored at the top of the iteration;
# timing-only: rows stored into an L2-resident region, only one of the two row kinds stored)
mkdir -p gpurun_out
L=$PWD/modulated_deform_conv_amd
{
MDCONV_LIB=$L/libmdconv_hip_f2t.so python tools/b1_timing.py --fwd2 cfg5 cfg3 2>&1 | grep -v amdgpu.ids
MDCONV_LIB=$L/libmdconv_hip_b3t.so python tools/b1_timing.py --bwd3 cfg5 2>&1 | grep -v amdgpu.ids
MDCONV_LIB=$L/libmdconv_hip_b1t.so python tools/b1_timing.py cfg4 cfg2 2>&1 | grep -v amdgpu.ids
MDCONV_LIB=$L/libmdconv_hip_b2t.so python tools/b1_timing.py --gemm2 cfg4 2>&1 | grep -v amdgpu.ids
for i in 1 2; do
python tools/exp.py cfg4 --label default --steps 20 2>&1 | grep -v amdgpu.ids
for v in b1ng b1ns b1ngns; do
MDCONV_LIB=$L/libmdconv_hip_$v.so python tools/exp.py cfg4 --label $v --steps 20 2>&1 | grep -v amdgpu.ids
done
MDCONV_BWD_FORK=0 python tools/exp.py cfg4 --label default-nofork --steps 20 2>&1 | grep -v amdgpu.ids
python tools/exp.py cfg5 --label default --steps 20 2>&1 | grep -v amdgpu.ids
for v in b3gq b3l2 b3gqo b3cqo; do
MDCONV_LIB=$L/libmdconv_hip_$v.so python tools/exp.py cfg5 --label $v --steps 20 2>&1 | grep -v amdgpu.ids
done
done
} > gpurun_out/ab_r05_19.txt 2>&1
cat gpurun_out/ab_r05_19.txt
