# 3-D conv groups with C_in / G not a multiple of 64: native NCHW kernels (MDCONV_PAD_CHANNELS=0) vs padded per group to 64 (default rule)
S="m3:f32:B2:C200:O64:8x20x20:g2 m3:f32:B2:C100:O16:8x20x20:g2 m3:f32:B2:C72:O40:8x20x20:g2 m3:f32:B2:C264:O64:8x20x20:g2 m3:f32:B2:C200:O64:8x20x20:g4 m3:f32:B2:C64:O64:8x20x20:g2 m3:f32:B2:C96:O96:8x28x28:g2 m3:f32:B4:C64:O128:8x14x14:g4"
for v in 0 x 0 x; do
  if [ $v = x ]; then unset MDCONV_PAD_CHANNELS; else export MDCONV_PAD_CHANNELS=$v; fi
  echo "=== $v"; python tools/prof_shape.py $S --n 20 2>&1 | grep " ms "
done
unset MDCONV_PAD_CHANNELS
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dgplan_forced.py -m gpu -x -q 2>&1 | tail -3
