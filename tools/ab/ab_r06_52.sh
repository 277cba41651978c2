# thresholds of the channel-padding rule: 3-D below 2048 pixels, 2-D wide layers at 25k-50k pixels (0 = native NCHW kernels, 1 = padded wherever eligible)
S="m3:f32:B4:C136:O64:4x7x7 m3:f32:B4:C200:O64:4x7x7 m3:f32:B4:C264:O64:4x7x7 m3:f32:B4:C40:O64:4x7x7 m3:f32:B2:C200:O64:4x14x14 m3:f32:B2:C136:O64:4x14x14 m2:f32:B4:C72:O64:112x112 m2:f32:B4:C200:O64:112x112 m2:f32:B4:C96:O96:112x112 m2:f32:B12:C96:O96:56x56 m2:f32:B12:C200:O64:56x56 m2:f32:B8:C200:O64:56x56 m2:f32:B8:C72:O64:56x56 m2:f32:B16:C160:O160:56x56"
for v in 0 1 0 1; do
  echo "=== MDCONV_PAD_CHANNELS=$v"
  MDCONV_PAD_CHANNELS=$v python tools/prof_shape.py $S --n 20 2>&1 | grep " ms "
done
