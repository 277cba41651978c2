# fewer than 16 input or output channels: shape-generic kernels (0) vs padded onto the matrix kernels (2; 2-D input channels to 16 or 64)
S3="m3:f32:B2:C8:O8:16x32x32 d3:f32:B2:C4:O8:16x32x32 m3:f32:B2:C12:O12:8x28x28 m3:f32:B2:C16:O8:8x28x28 m3:f32:B2:C64:O8:8x28x28 m3:f32:B1:C8:O8:8x16x16 m3:f32:B1:C4:O4:4x8x8 d3:f32:B1:C8:O8:4x14x14 m3:f32:B8:C3:O16:8x28x28"
S2="m2:f32:B8:C8:O8:112x112 m2:f32:B8:C12:O12:56x56 m2:f32:B8:C3:O16:112x112 m2:f32:B16:C64:O8:56x56 m2:f32:B4:C8:O8:28x28 d2:f32:B1:C4:O4:8x8 m2:f32:B2:C8:O8:56x56 m2:f32:B1:C8:O8:56x56"
echo "=== 0"; MDCONV_QUIET=1 MDCONV_PAD_CHANNELS=0 python tools/prof_shape.py $S3 $S2 --n 20 2>&1 | grep " ms "
echo "=== 2 (2-D C -> 64)"; MDCONV_QUIET=1 MDCONV_PAD_CHANNELS=2 python tools/prof_shape.py $S3 $S2 --n 20 2>&1 | grep " ms "
echo "=== 2 (2-D C -> 16)"; MDCONV_QUIET=1 MDCONV_PAD_CHANNELS=2 MDCONV_PAD_C2D=16 python tools/prof_shape.py $S2 --n 20 2>&1 | grep " ms "
MDCONV_PAD_CHANNELS=2 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fp32_auto_path" 2>&1 | tail -3
