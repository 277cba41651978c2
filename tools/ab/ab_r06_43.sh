for v in 0 1 x; do
  if [ $v = x ]; then unset MDCONV_PAD_CHANNELS; else export MDCONV_PAD_CHANNELS=$v; fi
  echo "== MDCONV_PAD_CHANNELS=$v"
  python tools/prof_shape.py m3:f32:B2:C160:O160:4x14x14 m3:f32:B2:C160:O160:4x14x14 m3:f32:B2:C192:O64:4x14x14 --n 30 2>&1 | grep " ms "
done
