# r05 call 5: fused prep launch (pack_wq + zero + layout pass) and grad_bias beside GEMM-2 on the forked stream
# "before" = libmdconv_hip_b3old.so, built from the tree before these two changes (its hp_bwd3 difference is not on cfg2 / cfg4)
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_modules.py tests/test_gpu_dist.py tests/test_gpu_ops.py tests/test_gpu_cl_forced.py -m gpu -q -x 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_fullshape_oracle.py -m gpu -q -x -k "cfg2 or cfg4" 2>&1 | tail -4
OLD=$PWD/modulated_deform_conv_amd/libmdconv_hip_b3old.so
for i in 1 2; do
python tools/exp.py cfg2 cfg2:4 cfg4 --label new --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_BWD_PREP=0 python tools/exp.py cfg2 cfg2:4 cfg4 --label prep-3-launches --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_LIB=$OLD python tools/exp.py cfg2 cfg2:4 cfg4 --label before --steps 20 2>&1 | grep -v amdgpu.ids
done
echo "## graph replays"
python tools/bench_configs.py cfg2 cfg2:16 cfg2:8 cfg2:4 --graph 2>&1 | grep -v amdgpu.ids
MDCONV_LIB=$OLD python tools/bench_configs.py cfg2 cfg2:4 --graph 2>&1 | grep -v amdgpu.ids
} > gpurun_out/ab_r05_5.txt 2>&1
cat gpurun_out/ab_r05_5.txt
