# tiny deformable groups: group-padded native 16-bit kernels (default) vs the route they took before (MDCONV_HP=0: fp32 kernels / shape-generic kernels on fp32 copies)
S="m2:f16:B8:C16:O16:56x56:dg4 m2:f16:B8:C8:O16:56x56:dg2 m2:f16:B8:C24:O24:56x56:dg3 m2:f16:B8:C4:O8:112x112:dg4 m2:f16:B8:C32:O32:56x56:dg8 m3:f16:B2:C16:O16:8x28x28:dg4 m2:f16:B8:C40:O40:56x56:dg5 m2:f32:B8:C16:O16:56x56:dg4 m2:f32:B8:C8:O16:56x56:dg2"
for v in 1 0; do
  echo "=== MDCONV_HP=$v"
  MDCONV_HP=$v MDCONV_QUIET=1 python tools/prof_shape.py $S --n 20 2>&1 | grep -v amdgpu.ids
done
echo "=== MDCONV_DG_PLAN=split (fp32 lines: the route before)"
MDCONV_DG_PLAN=split MDCONV_QUIET=1 python tools/prof_shape.py m2:f32:B8:C16:O16:56x56:dg4 m2:f32:B8:C8:O16:56x56:dg2 --n 20 2>&1 | grep -v amdgpu.ids
