mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_hp.py tests/test_gpu_hp_forced.py "tests/test_gpu_parity.py::test_fp32_non_finite_border_pixel_is_not_read" -m gpu -q 2>&1 | tail -40) > gpurun_out/t5.log 2>&1
(timeout 600 python -m pytest tests/test_gpu_fullshape_oracle.py -m gpu -q -k "cfg5 or cfg3" 2>&1 | tail -15) > gpurun_out/t5b.log 2>&1
python tools/bench_configs.py cfg5 cfg3 > gpurun_out/b5.txt 2>&1
python - >> gpurun_out/b5.txt 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench
for n in ("cfg3", "cfg5"):
    r = bench.time_other_config(n, "cuda")
    print(n, r["fwd_ms"], r["bwd_ms"], r["kernels_ms"])
PY
MDCONV_HP_BWD=2 python tools/bench_configs.py cfg5 >> gpurun_out/b5.txt 2>&1
tail -12 gpurun_out/t5.log; tail -5 gpurun_out/t5b.log; cat gpurun_out/b5.txt
