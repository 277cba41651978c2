# scatter-list fill: all taps of a 256-pixel block in one workgroup (MDCONV_HP_FILL=1) vs tap-major order (0)
for v in 0 1 0 1; do
  echo "=== MDCONV_HP_FILL=$v"
  MDCONV_HP_FILL=$v python tools/bench_configs.py cfg3 cfg5 2>&1 | grep "^cfg"
done
for v in 0 1; do
  echo "=== kernels MDCONV_HP_FILL=$v"
  MDCONV_HP_FILL=$v bash tools/prof_cfg.sh cfg3 cfg5 2>&1 | grep "== cfg\|csr_fill\|col2im_sums\|csr_scan"
done
timeout 600 python -m pytest tests/test_gpu_hp.py -m gpu -x -q -k "test_hp_fp16" 2>&1 | tail -2
MDCONV_HP_FILL=1 timeout 600 python -m pytest tests/test_gpu_hp.py -m gpu -x -q -k "test_hp_fp16 or test_hp_bf16" 2>&1 | tail -2
