#!/bin/bash
# usage: tools/pmc_cfg.sh "<counters space separated>" <cfg> [kernel-name-prefix]
# One rocprofv3 --pmc pass over tools/bench_configs.py <cfg>; prints per-kernel counter averages.
export TMPDIR=/tmp
ROOT=$(cd "$(dirname "$0")/.." && pwd)
D=$ROOT/gpurun_out/pmc_$2
rm -rf $D; mkdir -p $D
cd /tmp
timeout 250 rocprofv3 --pmc $1 --output-format csv -d $D -o p -- python $ROOT/tools/bench_configs.py $2 > $D/log.txt 2>&1
python3 - <<PY
import csv, collections, re
acc=collections.defaultdict(lambda: collections.defaultdict(list))
pref="${3-hp_}"
for r in csv.DictReader(open("$D/p_counter_collection.csv")):
    k=re.sub(r"\(.*","",r["Kernel_Name"].replace("void ","").replace("mdconv::(anonymous namespace)::","").replace("mdconv::",""))
    if k.startswith(pref): acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k[:44], " ".join("%s=%.4g"%(c.replace("SQ_","").replace("_CYCLES","_C"), sum(x)/len(x)) for c,x in sorted(v.items())))
PY
