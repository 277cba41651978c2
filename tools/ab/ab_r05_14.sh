# r05 call 14: scatter-list scan with its loads in flight (it runs beside a bandwidth-bound GEMM-2 on the forked stream)
# "before" = libmdconv_hip_prescan.so (the tree at dd9c1ec..c0 without the scan change)
mkdir -p gpurun_out
{
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hp.py tests/test_known_answers.py -m gpu -q -x 2>&1 | tail -3
OLD=$PWD/modulated_deform_conv_amd/libmdconv_hip_prescan.so
for i in 1 2; do
python tools/exp.py cfg5 cfg4 cfg3 cfg2 --label scan-new --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_LIB=$OLD python tools/exp.py cfg5 cfg4 cfg3 cfg2 --label before --steps 20 2>&1 | grep -v amdgpu.ids
done
bash tools/prof_cfg.sh cfg5 2>&1 | grep -i "scan\|fill\|gemm2\|sums\|== "
} > gpurun_out/ab_r05_14.txt 2>&1
cat gpurun_out/ab_r05_14.txt
