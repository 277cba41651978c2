#!/bin/bash
# usage: tools/pmc_kernel.sh "<counters space separated>" <out-tag> [bench args]
# Collects the counters (one rocprofv3 --pmc pass) for one bench.py run and prints per-kernel averages.
export TMPDIR=/tmp
ROOT=$(cd "$(dirname "$0")/.." && pwd)
D=$ROOT/gpurun_out/pmc_$2
mkdir -p $D
cd /tmp
timeout 250 rocprofv3 --pmc $1 --output-format csv -d $D -o p -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --sustain-s 0 > /dev/null 2>&1
python3 - <<PY
import csv, collections, re
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$D/p_counter_collection.csv")):
    k=re.sub(r"\(.*","",r["Kernel_Name"].replace("void ","").replace("mdconv::(anonymous namespace)::",""))
    if k.startswith(("mfma_","col2im")): acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k[:50], {c: "%.4g"%(sum(x)/len(x)) for c,x in v.items()})
PY
