# r05 call 1: parity of the hygiene batch, sub-batch stream pipeline (row g1), grid plans, baselines
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hp.py tests/test_gpu_fullshape_oracle.py tests/test_gpu_fullsize.py tests/test_gpu_cl_forced.py -m gpu -q -x 2>&1 | tail -8
echo "## sub-batch pipeline (tools/subbatch_pipeline.py)"
timeout 300 python tools/subbatch_pipeline.py --steps 20 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/subbatch_pipeline.py --steps 20 --graph 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/subbatch_pipeline.py --steps 20 --halves 4 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/subbatch_pipeline.py --steps 20 --halves 4 --graph 2>&1 | grep -v amdgpu.ids
echo "## plans"
MDCONV_DEBUG_PLAN=1 timeout 300 python tools/bench_configs.py cfg2 cfg2:4 cfg3 cfg4 cfg5 2>&1 | grep -v amdgpu.ids | sort | uniq -c | sort -rn | head -60
echo "## baselines"
timeout 300 python tools/exp.py cfg2 cfg3 cfg4 cfg5 cfg2:4 --label base --steps 20 2>&1 | grep -v amdgpu.ids
} > gpurun_out/ab_r05_1.txt 2>&1
cat gpurun_out/ab_r05_1.txt
