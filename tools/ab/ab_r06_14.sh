# few-tile shapes: pixel-stationary hp_bwd3 (one workgroup walks all taps) vs tap-stationary hp_bwd2 (MDCONV_HP_BWD=2)
S="m3:f16:B4:C256:O256:4x7x7:dg1 m3:f16:B4:C128:O128:4x14x14:dg1 m3:f16:B2:C64:O64:8x28x28:dg1 m3:f16:B4:C64:O128:8x14x14:dg1 m2:f16:B16:C256:O256:14x14:dg1 m2:f16:B16:C256:O256:7x7:dg1 m2:f16:B2:C256:O256:28x28:dg1 m2:f16:B16:C128:O128:28x28:dg1 m2:f16:B4:C128:O128:14x14:dg1 m2:f16:B16:C64:O64:14x14:dg1 m3:f16:B1:C128:O128:4x14x14:dg1 m3:f16:B8:C128:O128:4x14x14:dg1"
echo "== default"; python tools/prof_shape.py $S 2>&1 | grep -v amdgpu.ids
echo "== MDCONV_HP_BWD=2"; MDCONV_HP_BWD=2 python tools/prof_shape.py $S 2>&1 | grep -v amdgpu.ids
