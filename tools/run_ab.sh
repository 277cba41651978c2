mkdir -p gpurun_out
rm -f gpurun_out/b12.txt
L=$PWD/modulated_deform_conv_amd/libmdconv_hip_fwdpk.so
for v in default fwdpk; do
LL=""; [ $v != default ] && LL=$L
echo "== $v" >> gpurun_out/b12.txt
MDCONV_LIB=$LL python - >> gpurun_out/b12.txt 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench
for n in ("cfg3", "cfg5"):
    r = bench.time_other_config(n, "cuda")
    print(n, r["fwd_ms"], r["bwd_ms"], r["kernels_ms"])
PY
done
(MDCONV_LIB=$L timeout 600 python -m pytest tests/test_gpu_hp.py -m gpu -q 2>&1 | tail -12) >> gpurun_out/b12.txt 2>&1
(MDCONV_LIB=$L timeout 600 python -m pytest tests/test_gpu_fullshape_oracle.py -m gpu -q -k "cfg5 or cfg3" 2>&1 | tail -8) >> gpurun_out/b12.txt 2>&1
cat gpurun_out/b12.txt
