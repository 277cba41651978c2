# r05 call 16: instruction mix of a three-term bf16 split in GEMM-1's main loop (ABL_B1_BF16X3: TIMING ONLY, results wrong)
mkdir -p gpurun_out
{
for i in 1 2 3; do
python tools/exp.py cfg2 cfg2:4 --label fp32-exact --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_LIB=$PWD/modulated_deform_conv_amd/libmdconv_hip_bf16x3.so python tools/exp.py cfg2 cfg2:4 --label bf16x3-mix --steps 20 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/ab_r05_16.txt 2>&1
cat gpurun_out/ab_r05_16.txt
