# 16-bit backward, one deformable group, 96 / 160 / 192 / 224 channels: tap-stationary kernels (MDCONV_HP_PADC=0) vs padded to 128 / 256 on hp_bwd3 (1)
S="m2:f16:B8:C200:O256:40x40 m2:f16:B8:C96:O96:40x40 m2:f16:B8:C96:O96:56x56 m2:f16:B32:C96:O96:56x56 m2:f16:B8:C192:O192:56x56 m2:f16:B8:C160:O64:56x56 m2:f16:B16:C192:O192:28x28 m2:f16:B8:C224:O256:56x56 m3:f16:B2:C72:O64:8x28x28 m3:f16:B8:C96:O96:8x28x28 m3:f16:B2:C160:O160:8x28x28 m2:bf16:B8:C192:O256:56x56"
for v in 0 1 0 1; do
  echo "=== MDCONV_HP_PADC=$v"
  MDCONV_HP_PADC=$v python tools/prof_shape.py $S --n 20 2>&1 | grep " ms "
done
