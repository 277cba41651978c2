# r05 call 28: HIP events without the system-scope fence (hipEventDisableSystemFence).
#   evA = the library's TIMING events only (profile_mark: the per-kernel HIP events bench.py reads inside its timed region)
#   evB = evA + the dependency events (fork / join / bias of the forked backward, weights-ready)
# variants built from a temporary patch of mfma_kernels.hip / mdconv_api.hip (creation flags only), loaded through MDCONV_LIB
mkdir -p gpurun_out
P=$PWD/modulated_deform_conv_amd
{
for rep in 1 2; do
for v in default evA evB; do
  if [ $v = default ]; then unset MDCONV_LIB; else export MDCONV_LIB=$P/libmdconv_hip_$v.so; fi
  python tools/exp.py cfg2 cfg2:4 cfg3 cfg4 cfg5 --label $v 2>&1 | grep -v amdgpu.ids
done
done
for rep in 1 2; do
for v in default evA evB; do
  if [ $v = default ]; then unset MDCONV_LIB; else export MDCONV_LIB=$P/libmdconv_hip_$v.so; fi
  echo "bench.py $v: $(python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["ms_per_step"], j["ms_per_step_median"], j["sustained_ms_per_step"], j["kernels_ms"])')"
done
done
} > gpurun_out/ab_r05_28.txt 2>&1
cat gpurun_out/ab_r05_28.txt
