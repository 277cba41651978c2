# fp32 split backward: up to 4 deformable-group slices side by side (own streams, own workspace regions) vs one after the other
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_workspace_guard.py tests/test_gpu_modules.py tests/test_gpu_fuzz.py tests/test_gpu_cl_forced.py tests/test_gpu_concurrency.py -m gpu -x -q 2>&1 | tail -4
S="m2:f32:B16:C64:O64:56x56:dg4 m2:f32:B16:C128:O128:28x28:dg4 m2:f32:B8:C64:O256:56x56:dg4 m2:f32:B8:C96:O96:40x40:dg4 m2:f32:B8:C192:O192:20x20:dg4 m2:f32:B8:C320:O320:10x10:dg4 m2:f16:B8:C192:O192:20x20:dg4 m2:f32:B16:C64:O64:56x56:dg1"
echo "== lanes"; python tools/prof_shape.py $S 2>&1 | grep -v amdgpu.ids
echo "== MDCONV_SLICE_LANES=0"; MDCONV_SLICE_LANES=0 python tools/prof_shape.py $S 2>&1 | grep -v amdgpu.ids
