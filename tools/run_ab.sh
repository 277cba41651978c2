mkdir -p gpurun_out
rm -f gpurun_out/b18.txt
for v in default nont default nont; do
L=""; [ $v != default ] && L=$PWD/modulated_deform_conv_amd/libmdconv_hip_$v.so
echo "== $v" >> gpurun_out/b18.txt
MDCONV_LIB=$L python - >> gpurun_out/b18.txt 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench
for n in ("cfg5",):
    r = bench.time_other_config(n, "cuda")
    print(n, r["fwd_ms"], r["bwd_ms"], r["kernels_ms"])
PY
done
(timeout 600 python -m pytest tests/test_gpu_hp.py -m gpu -q -x 2>&1 | tail -3) >> gpurun_out/b18.txt 2>&1
cat gpurun_out/b18.txt
