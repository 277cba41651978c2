# r05 call 18: where hp_fwd2 / hp_bwd3 spend their time -- timing-only ablation builds (results wrong):
# no scattered gathers (every corner from one cache-resident row), one corner interpolated instead of 2^ND, no matrix phase,
# no row stores
mkdir -p gpurun_out
L=$PWD/modulated_deform_conv_amd
{
for i in 1 2; do
python tools/exp.py cfg5 cfg3 --label default --steps 20 2>&1 | grep -v amdgpu.ids
for v in f2ng f2ni f2nm f2ngni b3ng b3ns b3ngns; do
MDCONV_LIB=$L/libmdconv_hip_$v.so python tools/exp.py cfg5 cfg3 --label $v --steps 20 2>&1 | grep -v amdgpu.ids
done
done
} > gpurun_out/ab_r05_18.txt 2>&1
cat gpurun_out/ab_r05_18.txt
