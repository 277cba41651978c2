D=modulated_deform_conv_amd
for v in x_both x_nolds x_nocommit nogwb; do
  MDCONV_FWD_TAIL=0 MDCONV_LIB=$PWD/$D/libmdconv_hip_fa_$v.so python tools/exp.py cfg2 --label abl-$v --steps 10 2>&1 | grep -v amdgpu.ids
done
