# kernel breakdown of narrow fp32 3-D layers
cd /tmp && export TMPDIR=/tmp
for s in m3:f32:B2:C16:O16:16x32x32 m3:f32:B2:C32:O32:8x28x28 m3:f32:B2:C64:O64:8x28x28; do
    echo "=== kernels $s"
    rm -rf /tmp/prof_ab; (cd /root/repo && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -- python tools/prof_shape.py $s --n 20 2>&1 | grep " ms ")
    f=$(find /tmp/prof_ab -name '*kernel_stats.csv' | head -1)
    python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    if 'at::native' in r['Name']: continue
    print(f"{r['Name'].replace('mdconv::','').replace('(anonymous namespace)::','')[:80]:80s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
done
