# r05 call 26: the channels-last fp32 forward with the NCHW forward's tap-range tail plan -- parity, guarded workspaces, small
# forwards (graph replay) with and without the split, cfg4
mkdir -p gpurun_out
{
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_workspace_guard.py tests/test_gpu_cl_forced.py tests/test_gpu_fuzz.py tests/test_gpu_fullshape_oracle.py tests/test_gpu_fullsize.py tests/test_gpu_modules.py -m gpu -q -x 2>&1 | tail -4
echo "## tail plan"; python tools/small_fwd3d.py 2>&1 | grep -v amdgpu.ids
echo "## MDCONV_FWD_TAIL=0"; MDCONV_FWD_TAIL=0 python tools/small_fwd3d.py 2>&1 | grep -v amdgpu.ids
for i in 1 2; do
python tools/exp.py cfg4 --label tail-plan --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_FWD_TAIL=0 python tools/exp.py cfg4 --label no-split --steps 20 2>&1 | grep -v amdgpu.ids
done
MDCONV_DEBUG_PLAN=1 python tools/exp.py cfg4 --steps 2 2>&1 | grep "forward plan" | sort | uniq -c
} > gpurun_out/ab_r05_26.txt 2>&1
cat gpurun_out/ab_r05_26.txt
