# hp_bwd3 with A fragments from global memory (W^T slab > 48 KB): parity + the shapes it moves
timeout 900 python -m pytest tests/test_gpu_hp.py tests/test_gpu_hp_forced.py -m gpu -x -q 2>&1 | tail -8
python tools/prof_shape.py m2:f16:B8:C256:O256:56x56:dg1 m2:f16:B32:C256:O256:56x56:dg1 m2:f32:B8:C256:O256:56x56:dg1 m2:f16:B8:C256:O128:56x56:dg1 m2:f16:B8:C128:O256:56x56:dg1 m3:f16:B2:C256:O256:8x28x28:dg1 m2:bf16:B8:C256:O256:56x56:dg1 2>&1 | grep -v amdgpu.ids
export TMPDIR=/tmp; ROOT=$PWD; D=$ROOT/gpurun_out/r06b; rm -rf $D; mkdir -p $D
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o p -- python $ROOT/tools/prof_shape.py m2:f16:B8:C256:O256:56x56:dg1 > $D/log.txt 2>&1)
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$D/p_kernel_stats.csv")))
for r in rows[:14]:
    n=r["Name"].replace("void ","").replace("mdconv::(anonymous namespace)::","")[:90]
    print("  %-90s calls %5s avg_us %10.1f  %5.1f%%"%(n, r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
