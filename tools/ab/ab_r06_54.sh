# padded plan per conv group: parity, forced children, random shapes, anomaly sweep of the conv-group settings again
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dgplan_forced.py tests/test_gpu_workspace_guard.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -4
timeout 200 python tools/fuzz_more.py --seconds 150 --first 210000 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-500
MDCONV_QUIET=1 timeout 900 python tools/anomaly_sweep.py 2d groups f32 2>&1 | grep -v amdgpu.ids > gpurun_out/anom_2d_groups_f32_b.txt
MDCONV_QUIET=1 timeout 900 python tools/anomaly_sweep.py 3d groups f32 2>&1 | grep -v amdgpu.ids > gpurun_out/anom_3d_groups_f32_b.txt
grep -c "<<<" gpurun_out/anom_2d_groups_f32_b.txt gpurun_out/anom_3d_groups_f32_b.txt
