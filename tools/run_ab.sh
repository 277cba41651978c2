mkdir -p gpurun_out
rm -f gpurun_out/b14.txt
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_hp.py tests/test_gpu_hp_forced.py -m gpu -q -x 2>&1 | tail -6) >> gpurun_out/b14.txt 2>&1
(timeout 900 python -m pytest tests/test_gpu_fullshape_oracle.py -m gpu -q 2>&1 | tail -4) >> gpurun_out/b14.txt 2>&1
python - >> gpurun_out/b14.txt 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench
for n in ("cfg3", "cfg4", "cfg5"):
    r = bench.time_other_config(n, "cuda")
    print(n, r["fwd_ms"], r["bwd_ms"], r["kernels_ms"])
PY
MDCONV_BWD_FORK=0 python - >> gpurun_out/b14.txt 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench
r = bench.time_other_config("cfg4", "cuda")
print("cfg4 nofork", r["fwd_ms"], r["bwd_ms"], r["kernels_ms"])
PY
cat gpurun_out/b14.txt
