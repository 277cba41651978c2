# hp_bwd3 with deformable groups ((tap, group) gather units) + GEMM-2 range count for 8 output blocks: parity, then the shapes
timeout 1500 python -m pytest tests/test_gpu_hp.py tests/test_gpu_hp_forced.py tests/test_gpu_fuzz.py tests/test_gpu_workspace_guard.py -m gpu -x -q 2>&1 | tail -8
python tools/prof_shape.py m2:f16:B8:C256:O256:56x56:dg1 m2:f16:B8:C256:O256:56x56:dg4 m2:f16:B16:C64:O64:56x56:dg1 m2:f16:B16:C64:O64:56x56:dg4 m2:f16:B16:C128:O128:28x28:dg1 m2:f16:B16:C128:O128:28x28:dg4 m2:f16:B8:C64:O256:56x56:dg4 m2:f16:B8:C192:O192:20x20:dg4 2>&1 | grep -v amdgpu.ids
export TMPDIR=/tmp; ROOT=$PWD
for spec in m2:f16:B8:C256:O256:56x56:dg4 m2:f16:B16:C64:O64:56x56:dg4; do
D=$ROOT/gpurun_out/r06c; rm -rf $D; mkdir -p $D
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o p -- python $ROOT/tools/prof_shape.py $spec > $D/log.txt 2>&1)
echo "== $spec"
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$D/p_kernel_stats.csv")))
for r in rows[:12]:
    n=r["Name"].replace("void ","").replace("mdconv::(anonymous namespace)::","")[:90]
    print("  %-90s calls %5s avg_us %10.1f  %5.1f%%"%(n, r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
done
