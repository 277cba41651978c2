# r05 call 25: hp_gemm2 with a 128-register budget (four workgroups per CU; 28 registers spilled) against the shipped 168 / three
mkdir -p gpurun_out
L=$PWD/modulated_deform_conv_amd
{
for i in 1 2; do
python tools/exp.py cfg5 --label three-per-cu --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_LIB=$L/libmdconv_hip_g2b4.so python tools/exp.py cfg5 --label four-per-cu --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_BWD_FORK=0 python tools/exp.py cfg5 --label three-per-cu-nofork --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_BWD_FORK=0 MDCONV_LIB=$L/libmdconv_hip_g2b4.so python tools/exp.py cfg5 --label four-per-cu-nofork --steps 20 2>&1 | grep -v amdgpu.ids
done
MDCONV_DEBUG_PLAN=1 MDCONV_LIB=$L/libmdconv_hip_g2b4.so python tools/exp.py cfg5 --label plan --steps 2 2>&1 | grep "hp_gemm2" | sort | uniq -c
} > gpurun_out/ab_r05_25.txt 2>&1
cat gpurun_out/ab_r05_25.txt
