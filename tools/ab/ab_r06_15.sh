# crossover between hp_bwd3 (default) and hp_bwd2 (MDCONV_HP_BWD=2) over the pixel count
S=""
for B in 1 2 4 8 16 32; do S="$S m3:f16:B$B:C128:O128:4x14x14:dg1"; done
for B in 1 2 4 8 16 32; do S="$S m3:f16:B$B:C64:O64:8x28x28:dg1"; done
for B in 2 4 8 16 32 64; do S="$S m2:f16:B$B:C128:O128:28x28:dg1"; done
for B in 2 4 8 16 32 64; do S="$S m2:f16:B$B:C64:O64:56x56:dg1"; done
for B in 4 16 64; do S="$S m2:f16:B$B:C256:O64:28x28:dg1"; done
echo "== default"; python tools/prof_shape.py $S 2>&1 | grep -v amdgpu.ids
echo "== MDCONV_HP_BWD=2"; MDCONV_HP_BWD=2 python tools/prof_shape.py $S 2>&1 | grep -v amdgpu.ids
