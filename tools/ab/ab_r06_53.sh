# 2-D fp32, C_in >= 64 not a multiple of 64: NCHW kernels vs padded to the next 64 on the channels-last kernels, by pixel count
S="m2:f32:B8:C72:O64:56x56 m2:f32:B8:C96:O96:56x56 m2:f32:B8:C200:O64:56x56 m2:f32:B8:C160:O160:56x56 m2:f32:B12:C96:O96:56x56 m2:f32:B12:C200:O64:56x56 m2:f32:B4:C72:O64:112x112 m2:f32:B4:C96:O96:112x112 m2:f32:B4:C200:O64:112x112 m2:f32:B4:C160:O160:112x112 m2:f32:B32:C96:O96:56x56 m2:f32:B32:C200:O256:56x56 m2:f32:B8:C136:O128:112x112 d2:f32:B8:C96:O128:56x56"
for v in 0 1 0 1; do
  echo "=== $v"
  if [ $v = 1 ]; then export MDCONV_PAD_2D_N=8192; else unset MDCONV_PAD_2D_N; fi
  python tools/prof_shape.py $S --n 20 2>&1 | grep " ms "
done
