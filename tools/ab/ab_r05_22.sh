# r05 call 22: scatter lists of the fused 16-bit backward built on the side stream BESIDE the backward kernel
# (count -> scan -> fill); `before` = libmdconv_hip_pre.so (the tree before calls 21-22), `nofork` = MDCONV_BWD_FORK=0
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
L=$PWD/modulated_deform_conv_amd
{
timeout 1500 python -m pytest tests/test_gpu_hp.py tests/test_gpu_hp_forced.py tests/test_gpu_fuzz.py tests/test_gpu_modules.py tests/test_gpu_fullshape_oracle.py tests/test_analytic_pins.py -m gpu -q -x 2>&1 | tail -5
for i in 1 2 3; do
python tools/exp.py cfg3 --label new --steps 50 2>&1 | grep -v amdgpu.ids
MDCONV_LIB=$L/libmdconv_hip_pre.so python tools/exp.py cfg3 --label before --steps 50 2>&1 | grep -v amdgpu.ids
MDCONV_BWD_FORK=0 python tools/exp.py cfg3 --label new-nofork --steps 50 2>&1 | grep -v amdgpu.ids
done
echo "## graph replays"
python tools/bench_configs.py cfg3 --graph 2>&1 | grep -v amdgpu.ids | grep graph
MDCONV_LIB=$L/libmdconv_hip_pre.so python tools/bench_configs.py cfg3 --graph 2>&1 | grep -v amdgpu.ids | grep graph
D=$ROOT/gpurun_out/trace_cfg3
rm -rf $D; mkdir -p $D
(cd /tmp && timeout 280 rocprofv3 --kernel-trace --output-format csv -d $D -o p -- python $ROOT/tools/exp.py cfg3 --steps 6 > $D/log.txt 2>&1)
} > gpurun_out/ab_r05_22.txt 2>&1
cat gpurun_out/ab_r05_22.txt
