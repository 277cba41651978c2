# round-5 tree (worktree _r5) vs the current tree on the lines of the realistic-shape sweep that got slower, with kernel breakdowns
S="m2:f16:B8:C2048:O512:7x7:dg4 m2:f16:B8:C2048:O512:7x7:dg1 m2:f16:B16:C512:O512:7x7:dg4 m2:f16:B8:C1024:O1024:7x7:dg4 m2:f16:B8:C1024:O1024:7x7:dg1 m3:f16:B4:C256:O256:4x7x7:dg1"
cd /tmp && export TMPDIR=/tmp
for tree in /root/repo/_r5 /root/repo; do
  echo "=== $tree"
  (cd $tree && python tools/prof_shape.py $S --n 20 2>&1 | grep -v amdgpu.ids)
done
for s in m2:f16:B8:C2048:O512:7x7:dg4 m2:f16:B16:C512:O512:7x7:dg4 m3:f16:B4:C256:O256:4x7x7:dg1; do
  for tree in /root/repo/_r5 /root/repo; do
    echo "=== kernels $s $tree"
    rm -rf /tmp/prof_ab; (cd $tree && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -- python tools/prof_shape.py $s --n 20 > /dev/null 2>&1)
    f=$(find /tmp/prof_ab -name '*kernel_stats.csv' | head -1)
    python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
  done
done
