# r05 call 23: what GEMM-1's tap state costs -- timing-only ablations: its coefficient arithmetic done twice (b1s2),
# no counting atomics / tap-table entry (b1nt: GEMM-2 and the gather then read garbage, only mfma_bwd_data's time counts)
mkdir -p gpurun_out
L=$PWD/modulated_deform_conv_amd
{
for i in 1 2; do
python tools/exp.py cfg2 cfg4 --label default --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_LIB=$L/libmdconv_hip_b1s2.so python tools/exp.py cfg2 cfg4 --label state-twice --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_LIB=$L/libmdconv_hip_b1nt.so timeout 120 python tools/exp.py cfg2 cfg4 --label no-table --steps 20 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/ab_r05_23.txt 2>&1
cat gpurun_out/ab_r05_23.txt
