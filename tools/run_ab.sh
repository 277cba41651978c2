mkdir -p gpurun_out
rm -f gpurun_out/b13.txt
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_cl_forced.py -m gpu -q -x 2>&1 | tail -6) >> gpurun_out/b13.txt 2>&1
(timeout 600 python -m pytest tests/test_gpu_fullshape_oracle.py -m gpu -q -k "cfg4" 2>&1 | tail -4) >> gpurun_out/b13.txt 2>&1
for m in 2 1; do
echo "== C2I3D=$m" >> gpurun_out/b13.txt
MDCONV_C2I3D=$m python - >> gpurun_out/b13.txt 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench
r = bench.time_other_config("cfg4", "cuda")
print("cfg4", r["fwd_ms"], r["bwd_ms"], r["kernels_ms"])
PY
MDCONV_BWD_FORK=0 MDCONV_C2I3D=$m python - >> gpurun_out/b13.txt 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench
r = bench.time_other_config("cfg4", "cuda")
print("cfg4 nofork", r["fwd_ms"], r["bwd_ms"], r["kernels_ms"])
PY
done
bash tools/prof_cfg.sh cfg4 >> gpurun_out/b13.txt 2>&1
cat gpurun_out/b13.txt
