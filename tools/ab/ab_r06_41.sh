# channel padding to 64: 16-channel 3-D layers (bit 2) and 2-D layers of >= 8192 pixels (bit 4)
S="m3:f32:B2:C16:O16:16x32x32 d3:f32:B2:C16:O32:8x28x28 m3:f32:B2:C24:O24:8x28x28 m2:f32:B8:C96:O96:40x40 m2:f32:B8:C32:O32:112x112 m2:f32:B16:C48:O48:56x56 m2:f32:B16:C160:O160:28x28 m2:f32:B8:C320:O320:40x40 m2:f32:B16:C32:O64:56x56 d2:f32:B8:C96:O128:56x56"
for v in 0 7 0 7; do
  echo "=== MDCONV_PAD3D=$v"
  MDCONV_PAD3D=$v python tools/prof_shape.py $S --n 20 2>&1 | grep -v amdgpu.ids
done
