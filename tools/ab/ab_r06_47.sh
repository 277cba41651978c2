# padded plan for everything the fp32 kernels do not tile (one conv group): parity, random shapes, anomaly sweep again
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dgplan_forced.py tests/test_gpu_workspace_guard.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -4
timeout 200 python tools/fuzz_more.py --seconds 150 --first 180000 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-500
timeout 200 python tools/fuzz_more.py --seconds 100 --first 181000 --wide 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-500
MDCONV_QUIET=1 timeout 900 python tools/anomaly_sweep.py 3d g1 f32 2>&1 | grep -v amdgpu.ids > gpurun_out/anom_3d_f32_b.txt
MDCONV_QUIET=1 timeout 900 python tools/anomaly_sweep.py 2d g1 f32 2>&1 | grep -v amdgpu.ids > gpurun_out/anom_2d_f32_b.txt
grep -c "<<<" gpurun_out/anom_3d_f32_b.txt gpurun_out/anom_2d_f32_b.txt
