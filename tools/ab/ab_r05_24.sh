# r05 call 24: the channels-last fp32 forward (cfg4) -- timing-only ablations: no corner loads, corner loads from one
# cache-resident row, no matrix instructions
mkdir -p gpurun_out
L=$PWD/modulated_deform_conv_amd
{
for i in 1 2; do
python tools/exp.py cfg4 --label default --steps 20 2>&1 | grep -v amdgpu.ids
for v in fcnl fcng fcnm fcnb fcnw fcna fcnlna fcall; do
MDCONV_LIB=$L/libmdconv_hip_$v.so python tools/exp.py cfg4 --label $v --steps 20 2>&1 | grep -v amdgpu.ids
done
done
} > gpurun_out/ab_r05_24.txt 2>&1
cat gpurun_out/ab_r05_24.txt
