# scatter lists built beside GEMM-1 (csr_count -> scan -> fill on the forked stream) vs the round-5 order (MDCONV_EARLY_CSR=0)
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_modules.py tests/test_gpu_concurrency.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2; do
python tools/exp.py cfg2 cfg2:4 cfg4 --label early-csr --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_EARLY_CSR=0 python tools/exp.py cfg2 cfg2:4 cfg4 --label r5-order --steps 20 2>&1 | grep -v amdgpu.ids
done
python tools/bench_configs.py cfg2 cfg2:16 cfg2:8 cfg2:4 --graph 2>&1 | grep -v amdgpu.ids
MDCONV_EARLY_CSR=0 python tools/bench_configs.py cfg2 cfg2:16 cfg2:8 cfg2:4 --graph 2>&1 | grep -v amdgpu.ids
