# 3-D fp32 layers whose channel count is not a multiple of 64: NCHW kernels (MDCONV_PAD3D=0) vs padded to 64 on the channels-last kernels (1)
S="m3:f32:B2:C32:O32:8x28x28 d3:f32:B2:C32:O32:8x28x28 m3:f32:B2:C48:O48:8x28x28 m3:f32:B2:C96:O96:8x28x28 m3:f32:B4:C96:O128:8x14x14 m3:f32:B2:C160:O160:4x14x14 m3:f32:B2:C32:O64:16x56x56 m3:f32:B8:C32:O32:4x7x7 m3:f32:B1:C40:O40:8x16x16 m3:f32:B2:C192:O64:4x14x14 m3:f16:B2:C320:O64:4x14x14"
for v in 0 1 0 1; do
  echo "=== MDCONV_PAD3D=$v"
  MDCONV_PAD3D=$v python tools/prof_shape.py $S --n 20 2>&1 | grep -v amdgpu.ids
done
