# r05 call 30: fp32 forward (mfma_fwd.hip), bias values loaded eight at a time ahead of their stores -- one-file variant fb2 against the tree
mkdir -p gpurun_out
P=$PWD/modulated_deform_conv_amd
{
for rep in 1 2 3; do
for v in default fb2; do
  if [ $v = default ]; then unset MDCONV_LIB; else export MDCONV_LIB=$P/libmdconv_hip_$v.so; fi
  python tools/exp.py cfg2 cfg2:4 --label $v 2>&1 | grep -v amdgpu.ids
done
done
} > gpurun_out/ab_r05_30.txt 2>&1
cat gpurun_out/ab_r05_30.txt
