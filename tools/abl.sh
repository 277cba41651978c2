for v in "$@"; do
  if [ "$v" = base ]; then lib=modulated_deform_conv_amd/libmdconv_hip.so; else lib=modulated_deform_conv_amd/libmdconv_hip_$v.so; fi
  echo "== $v"
  MDCONV_LIB=$PWD/$lib timeout 200 python bench.py --no-cpu-baseline --steps 10 --warmup 3 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernels_ms'])"
done
