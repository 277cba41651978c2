# kernel breakdown of the fp16 64-channel layer with 4 / 1 deformable groups (1.76x)
cd /tmp && export TMPDIR=/tmp
for s in m2:f16:B16:C64:O64:56x56:dg4 m2:f16:B16:C64:O64:56x56:dg1 m2:f32:B16:C128:O128:28x28:dg4 m2:f32:B16:C128:O128:28x28:dg1; do
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
