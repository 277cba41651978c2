# r05 call 13: cache policy of hp_bwd3's row stores (aux bits: 2 = nt (shipped), 16 = sc1, 18 = sc1 nt, 17 = sc0 sc1, 0 = plain)
mkdir -p gpurun_out
{
for i in 1 2; do
python tools/exp.py cfg5 --label nt-2 --steps 20 2>&1 | grep -v amdgpu.ids
for v in st16 st18 st17 st0; do
MDCONV_LIB=$PWD/modulated_deform_conv_amd/libmdconv_hip_$v.so python tools/exp.py cfg5 --label $v --steps 20 2>&1 | grep -v amdgpu.ids
done
done
} > gpurun_out/ab_r05_13.txt 2>&1
cat gpurun_out/ab_r05_13.txt
