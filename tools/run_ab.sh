mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_hp.py tests/test_gpu_hp_forced.py -m gpu -q -x 2>&1 | tail -15) > gpurun_out/t7.log 2>&1
(timeout 600 python -m pytest tests/test_gpu_fullshape_oracle.py -m gpu -q -k "cfg5 or cfg3" 2>&1 | tail -15) > gpurun_out/t7b.log 2>&1
rm -f gpurun_out/b7.txt
for m in 2 1; do
echo "== C2I=$m" >> gpurun_out/b7.txt
MDCONV_HP_C2I=$m python - >> gpurun_out/b7.txt 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench
for n in ("cfg3", "cfg5"):
    r = bench.time_other_config(n, "cuda")
    print(n, r["fwd_ms"], r["bwd_ms"], r["kernels_ms"])
PY
done
bash tools/prof_cfg.sh cfg5 cfg3 >> gpurun_out/b7.txt 2>&1
tail -4 gpurun_out/t7.log; tail -3 gpurun_out/t7b.log; cat gpurun_out/b7.txt
