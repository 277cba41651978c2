export TMPDIR=/tmp; ROOT=$PWD
for spec in m2:f16:B32:C256:O256:56x56:dg1 m2:f16:B16:C64:O64:56x56:dg4; do
D=$ROOT/gpurun_out/r06d; rm -rf $D; mkdir -p $D
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o p -- python $ROOT/tools/prof_shape.py $spec > $D/log.txt 2>&1)
grep -v amdgpu.ids $D/log.txt | tail -1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$D/p_kernel_stats.csv")))
for r in rows[:13]:
    n=r["Name"].replace("void ","").replace("mdconv::(anonymous namespace)::","")[:90]
    print("  %-90s calls %5s avg_us %10.1f  %5.1f%%"%(n, r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
done
