# cfg5: hp_bwd3 with A fragments from global memory (no slab: 45 KB of LDS) at 2 and at 3 waves per SIMD (184 B of scratch)
L=$PWD/modulated_deform_conv_amd
for i in 1 2; do
python tools/exp.py cfg5 --label default --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_LIB=$L/libmdconv_hip_wg2.so python tools/exp.py cfg5 --label wg-2waves --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_LIB=$L/libmdconv_hip_wg3.so python tools/exp.py cfg5 --label wg-3waves-spill --steps 20 2>&1 | grep -v amdgpu.ids
done
