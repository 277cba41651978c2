# r05 call 27: few-tile many-stage 16-bit forwards routed to the fp32 kernels (hp_forward_preferred) -- parity, sweep before / after
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_gpu_hp.py tests/test_gpu_hp_forced.py tests/test_gpu_fuzz.py tests/test_gpu_modules.py tests/test_gpu_workspace_guard.py tests/test_gpu_ops.py tests/test_analytic_pins.py -m gpu -q -x 2>&1 | tail -4
for args in "M2 f16 16 512 512 7 7 -- 1" "M2 f16 1 512 512 7 7 -- 1" "M2 f16 8 1024 1024 7 7 -- 1" "M3 f16 4 256 256 4 7 7 -- 1" "M3 f16 1 256 256 4 14 14 -- 1" "M2 f16 1 2048 512 7 7 -- 1"; do
python tools/why_slow.py $args 2>&1 | grep -v amdgpu.ids | tail -1
MDCONV_HP_FWD=2 python tools/why_slow.py $args 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/^/   native forward forced: /'
done
} > gpurun_out/ab_r05_27.txt 2>&1
cat gpurun_out/ab_r05_27.txt
