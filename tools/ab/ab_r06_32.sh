# fp32 deformable groups the kernels do not tile: ONE padded problem (MDCONV_DG_PLAN=pad) vs DG single-group slices (split)
S="m2:f32:B16:C64:O64:56x56:dg4 m2:f32:B16:C128:O128:28x28:dg4 m2:f32:B8:C64:O256:56x56:dg4 m2:f32:B8:C96:O96:40x40:dg4 m2:f32:B8:C192:O192:20x20:dg4 m2:f32:B8:C320:O320:10x10:dg4 m2:f32:B8:C32:O32:112x112:dg2 m2:f32:B16:C128:O128:28x28:dg2 m2:f32:B8:C256:O256:28x28:dg8 m3:f32:B2:C64:O64:8x28x28:dg2 m3:f32:B2:C64:O64:8x28x28:dg4 m2:f16:B8:C320:O320:10x10:dg4 m2:f32:B4:C16:O16:56x56:dg2"
for v in split pad; do
  echo "=== MDCONV_DG_PLAN=$v"
  MDCONV_DG_PLAN=$v python tools/prof_shape.py $S --n 20 2>&1 | grep -v amdgpu.ids
done
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_workspace_guard.py -m gpu -x -q 2>&1 | tail -4
MDCONV_DG_PLAN=pad timeout 200 python tools/fuzz_more.py --seconds 120 --first 95000 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-600
