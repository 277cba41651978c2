mkdir -p gpurun_out
SPLIT=$PWD/modulated_deform_conv_amd/libmdconv_hip_split.so
(timeout 600 python -m pytest tests/test_gpu_hp.py -m gpu -x -q 2>&1 | tail -5) > gpurun_out/t3_fused_brick.log 2>&1
(MDCONV_LIB=$SPLIT timeout 600 python -m pytest tests/test_gpu_hp.py -m gpu -x -q 2>&1 | tail -5) > gpurun_out/t3_split_brick.log 2>&1
for v in fused split; do for br in 1 0; do
  L=""; [ $v = split ] && L=$SPLIT
  echo "== $v brick=$br" >> gpurun_out/b3.txt
  MDCONV_LIB=$L MDCONV_HP_BRICK=$br python tools/bench_configs.py cfg3 cfg5 >> gpurun_out/b3.txt 2>&1
  MDCONV_LIB=$L MDCONV_HP_BRICK=$br python - >> gpurun_out/b3.txt 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench
for n in ("cfg3", "cfg5"):
    r = bench.time_other_config(n, "cuda")
    print(n, r["fwd_ms"], r["bwd_ms"], r["kernels_ms"])
PY
done; done
tail -3 gpurun_out/t3_fused_brick.log gpurun_out/t3_split_brick.log; cat gpurun_out/b3.txt
