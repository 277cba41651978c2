# r05 call 10: forward tail plan that fills the last (or only) dispatch round with mixed tap-range counts
# (MDCONV_FWD_TAIL=1 = the round-4 plan: uniform ranges, only behind a full round)
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_modules.py -m gpu -q -x 2>&1 | tail -4
MDCONV_DEBUG_PLAN=1 python tools/exp.py cfg2:4 cfg2:8 cfg2:16 cfg2 --label plan --steps 2 2>&1 | grep "forward plan" | sort | uniq -c
for i in 1 2; do
python tools/exp.py cfg2:4 cfg2:8 cfg2:16 cfg2 --label mixed-tail --steps 30 2>&1 | grep -v amdgpu.ids
MDCONV_FWD_TAIL=1 python tools/exp.py cfg2:4 cfg2:8 cfg2:16 cfg2 --label r4-tail --steps 30 2>&1 | grep -v amdgpu.ids
done
echo "## graph replays"
python tools/bench_configs.py cfg2 cfg2:16 cfg2:8 cfg2:4 --graph 2>&1 | grep -v amdgpu.ids
MDCONV_FWD_TAIL=1 python tools/bench_configs.py cfg2 cfg2:16 cfg2:8 cfg2:4 --graph 2>&1 | grep -v amdgpu.ids
} > gpurun_out/ab_r05_10.txt 2>&1
cat gpurun_out/ab_r05_10.txt
