# r05 call 15: hp_gemm2 made of MORE, shorter workgroups (the forked gather's first kernels wait 0.55 ms for a CU slot behind its
# 0.7-ms workgroups): workgroups per CU x 256 / 27 pixel ranges; 4 = the shipped sizing (37 ranges, 999 workgroups)
mkdir -p gpurun_out
{
V=$PWD/modulated_deform_conv_amd/libmdconv_hip_g2rw.so
for i in 1 2; do
for n in 4 8 12 16 24; do
MDCONV_LIB=$V MDCONV_HP_G2_SLOTS=$n python tools/exp.py cfg5 --label g2-slots-$n --steps 20 2>&1 | grep -v amdgpu.ids
done
done
MDCONV_LIB=$V MDCONV_HP_G2_SLOTS=12 bash tools/prof_cfg.sh cfg5 2>&1 | grep -i "scan\|fill\|gemm2\|sums\|reduce\|== "
} > gpurun_out/ab_r05_15.txt 2>&1
cat gpurun_out/ab_r05_15.txt
