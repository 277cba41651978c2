mkdir -p gpurun_out
rm -f gpurun_out/b17.txt
(timeout 900 python -m pytest tests/test_gpu_hp.py tests/test_gpu_hp_forced.py tests/test_gpu_modules.py -m gpu -q -x 2>&1 | tail -4) >> gpurun_out/b17.txt 2>&1
(timeout 900 python -m pytest tests/test_gpu_fullshape_oracle.py -m gpu -q -k "cfg3 or cfg5" 2>&1 | tail -3) >> gpurun_out/b17.txt 2>&1
for f in 1 0; do
echo "== fork=$f" >> gpurun_out/b17.txt
MDCONV_BWD_FORK=$f python - >> gpurun_out/b17.txt 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench
for n in ("cfg3", "cfg5"):
    r = bench.time_other_config(n, "cuda")
    print(n, r["fwd_ms"], r["bwd_ms"], r["kernels_ms"])
PY
done
cat gpurun_out/b17.txt
