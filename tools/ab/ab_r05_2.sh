# r05 call 2: pair-wise state building in hp_fwd2 / hp_bwd3 (A/B against one-file variants with the old build)
mkdir -p gpurun_out
{
timeout 1200 python -m pytest tests/test_gpu_hp.py tests/test_gpu_hp_forced.py tests/test_analytic_pins.py -m gpu -q -x 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_fullshape_oracle.py -m gpu -q -x -k "cfg3 or cfg5" 2>&1 | tail -4
for i in 1 2; do
python tools/exp.py cfg3 cfg5 --label pair --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_LIB=$PWD/modulated_deform_conv_amd/libmdconv_hip_f2old.so python tools/exp.py cfg3 cfg5 --label fwd2-old --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_LIB=$PWD/modulated_deform_conv_amd/libmdconv_hip_b3old.so python tools/exp.py cfg3 cfg5 --label bwd3-old --steps 20 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/ab_r05_2.txt 2>&1
cat gpurun_out/ab_r05_2.txt
