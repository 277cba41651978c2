# r05 call 21: single-pass counter scan, grad_bias stage 1 / split-K reduction with more loads in flight -- parity subset,
# A/B against the library before (libmdconv_hip_pre.so), kernel timelines
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
L=$PWD/modulated_deform_conv_amd
{
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hp.py tests/test_known_answers.py tests/test_gpu_fullshape_oracle.py tests/test_gpu_modules.py -m gpu -q -x 2>&1 | tail -5
for i in 1 2; do
python tools/exp.py cfg2 cfg2:4 cfg3 cfg4 cfg5 --label new --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_LIB=$L/libmdconv_hip_pre.so python tools/exp.py cfg2 cfg2:4 cfg3 cfg4 cfg5 --label before --steps 20 2>&1 | grep -v amdgpu.ids
done
echo "## graph replays"
python tools/bench_configs.py cfg2 cfg2:16 cfg2:8 cfg2:4 --graph 2>&1 | grep -v amdgpu.ids | grep graph
MDCONV_LIB=$L/libmdconv_hip_pre.so python tools/bench_configs.py cfg2 cfg2:16 cfg2:8 cfg2:4 --graph 2>&1 | grep -v amdgpu.ids | grep graph
for c in cfg2 cfg2:4 cfg5; do
  D=$ROOT/gpurun_out/trace_$c
  rm -rf $D; mkdir -p $D
  (cd /tmp && timeout 280 rocprofv3 --kernel-trace --output-format csv -d $D -o p -- python $ROOT/tools/exp.py $c --steps 6 > $D/log.txt 2>&1)
done
} > gpurun_out/ab_r05_21.txt 2>&1
cat gpurun_out/ab_r05_21.txt
