export TMPDIR=/tmp
ROOT=$PWD
mkdir -p gpurun_out/r06a
python bench.py --steps 20 --warmup 5 > gpurun_out/r06a/bench.json 2> gpurun_out/r06a/bench.err
prof() { # name spec...
  D=$ROOT/gpurun_out/r06a/prof_$1; rm -rf $D; mkdir -p $D; shift
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o p -- python $ROOT/tools/prof_shape.py "$@" > $D/log.txt 2>&1)
  cat $D/log.txt | grep -v amdgpu.ids
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("$D/p_kernel_stats.csv")))
for r in rows[:16]:
    n=r["Name"].replace("void ","").replace("mdconv::(anonymous namespace)::","")[:90]
    print("  %-90s calls %5s avg_us %10.1f  %5.1f%%"%(n, r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
}
{
prof a m2:f16:B8:C256:O256:56x56:dg1
prof b m2:f32:B16:C64:O64:56x56:dg4
prof c m2:f32:B16:C64:O64:56x56:dg1
prof d m2:f16:B16:C64:O64:56x56:dg4
prof e m2:f16:B16:C64:O64:56x56:dg1
prof f m2:f32:B16:C128:O128:28x28:dg4
prof g m2:f16:B8:C256:O256:56x56:dg4
} > gpurun_out/r06a/shapes.txt 2>&1
cat gpurun_out/r06a/shapes.txt
