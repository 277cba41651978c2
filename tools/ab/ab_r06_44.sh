# fewer than 16 channels (shape-generic kernels today) padded to 64?
S="m3:f32:B2:C8:O8:16x32x32 m3:f32:B2:C8:O16:8x28x28 d3:f32:B2:C4:O8:16x32x32 m3:f32:B2:C12:O12:8x28x28 m2:f32:B8:C8:O8:112x112 m2:f32:B8:C12:O12:56x56 m3:f32:B2:C16:O8:8x28x28"
for v in 0 2 0 2; do
  echo "=== MDCONV_PAD_CHANNELS=$v"
  MDCONV_QUIET=1 MDCONV_PAD_CHANNELS=$v python tools/prof_shape.py $S --n 20 2>&1 | grep " ms "
done
