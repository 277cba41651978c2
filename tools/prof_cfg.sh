#!/bin/bash
# usage: tools/prof_cfg.sh cfg4 [cfg5 ...] : rocprofv3 kernel stats of tools/bench_configs.py <cfg>
export TMPDIR=/tmp
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for c in "$@"; do
  D=$ROOT/gpurun_out/prof_$c
  rm -rf $D; mkdir -p $D
  (cd /tmp && timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o p -- python $ROOT/tools/bench_configs.py $c > $D/log.txt 2>&1)
  echo "== $c"; tail -1 $D/log.txt
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("$D/p_kernel_stats.csv")))
for r in rows[:14]:
    n=r["Name"].replace("void ","").replace("mdconv::(anonymous namespace)::","")[:70]
    print("%-70s calls %5s avg_us %10.1f  %5.1f%%"%(n, r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
done
