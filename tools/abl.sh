for ss in 1 0; do
  echo "== side stream $ss"
  MDCONV_SIDE_STREAM=$ss timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['kernels_ms'])"
done
