# r05 call 4: upper bound of an LDS-staged input window for the cfg3 forward (ABL_FWD_WINDOW: timing only)
mkdir -p gpurun_out
{
for i in 1 2 3; do
python tools/exp.py cfg3 --label default --steps 50 2>&1 | grep -v amdgpu.ids
MDCONV_LIB=$PWD/modulated_deform_conv_amd/libmdconv_hip_fwin.so python tools/exp.py cfg3 --label lds-window-bound --steps 50 2>&1 | grep -v amdgpu.ids
done
MDCONV_DEBUG_PLAN=1 MDCONV_LIB=$PWD/modulated_deform_conv_amd/libmdconv_hip_fwin.so python tools/exp.py cfg3 --label plan --steps 2 2>&1 | grep "hp_fwd2" | sort | uniq -c
} > gpurun_out/ab_r05_4.txt 2>&1
cat gpurun_out/ab_r05_4.txt
