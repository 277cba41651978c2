# kernel breakdowns, round-5 tree vs current, large-N small-C fp16 lines that the sweep shows 13-29 % slower
cd /tmp && export TMPDIR=/tmp
for s in m2:f16:B4:C64:O64:112x112:dg1 m2:f16:B8:C128:O128:56x56:dg2 m2:f16:B8:C32:O32:112x112:dg1; do
  for tree in /root/repo/_r5 /root/repo /root/repo/_r5 /root/repo; do
    (cd $tree && python tools/prof_shape.py $s --n 50 2>&1 | grep " ms " | sed "s|^|$(basename $tree) |")
  done
  for tree in /root/repo/_r5 /root/repo; do
    echo "=== kernels $s $tree"
    rm -rf /tmp/prof_ab; (cd $tree && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -- python tools/prof_shape.py $s --n 20 2>&1 | grep " ms ")
    f=$(find /tmp/prof_ab -name '*kernel_stats.csv' | head -1)
    python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    if 'at::native' in r['Name']: continue
    print(f"{r['Name'].replace('mdconv::','').replace('(anonymous namespace)::','')[:80]:80s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
  done
done
