# full GPU suite after the group-padded plans (16-bit native + fp32 padded problem), then the DG lines of the sweep
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python tools/prof_shape.py m2:f32:B16:C64:O64:56x56:dg4 m2:f32:B8:C96:O96:40x40:dg4 m2:f32:B8:C192:O192:20x20:dg4 m2:f32:B8:C320:O320:10x10:dg4 m2:f16:B8:C320:O320:10x10:dg4 m2:f32:B4:C16:O16:56x56:dg2 2>&1 | grep -v amdgpu.ids
timeout 200 python tools/fuzz_more.py --seconds 120 --first 96000 --pad 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-600
