# r05 call 3: 64-pixel-slab GEMM-2 for the narrow tiles (MDCONV_BW_SLAB=0 = the 16-pixel-chunk kernel)
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cl_forced.py tests/test_gpu_fuzz.py tests/test_analytic_pins.py -m gpu -q -x 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_fullshape_oracle.py tests/test_gpu_fullsize.py -m gpu -q -x -k "cfg4" 2>&1 | tail -4
for i in 1 2; do
python tools/exp.py cfg4 --label slab64 --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_BW_SLAB=0 python tools/exp.py cfg4 --label chunk16 --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_BWD_FORK=0 python tools/exp.py cfg4 --label slab64-nofork --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_BWD_FORK=0 MDCONV_BW_SLAB=0 python tools/exp.py cfg4 --label chunk16-nofork --steps 20 2>&1 | grep -v amdgpu.ids
done
MDCONV_DEBUG_PLAN=1 python tools/exp.py cfg4 --label plan --steps 2 2>&1 | grep "GEMM-2 plan" | sort | uniq -c
python tools/cl_sweep.py 2>&1 | grep -v amdgpu.ids | tail -30
} > gpurun_out/ab_r05_3.txt 2>&1
cat gpurun_out/ab_r05_3.txt
