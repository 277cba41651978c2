#!/bin/bash
# Run ON THE GPU BOX after tools/collect_profiles.sh <tag>: counter HBM traffic of one forward + backward STEP of every
# BASELINE.json configuration (separate FETCH_SIZE / WRITE_SIZE passes over tools/bench_configs.py <cfg>: 11 forward and
# 11 backward calls each) -> gpurun_out/prof_<tag>/steps_<cfg>.json, which tools/summarize_profile.py folds into
# profiles/<tag>_pmc_summary.json ("_steps") for bench.py's `traffic` fields.   usage: tools/collect_steps.sh r06
export TMPDIR=/tmp
ROOT=$(cd "$(dirname "$0")/.." && pwd)
D=$ROOT/gpurun_out/prof_$1
mkdir -p $D
for c in cfg2 cfg3 cfg4 cfg5; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $D/steps_${c}_$ctr
    (cd /tmp && timeout 280 rocprofv3 --pmc $ctr --output-format csv -d $D/steps_${c}_$ctr -o p -- python $ROOT/tools/bench_configs.py $c > /dev/null 2>&1)
  done
  python3 - <<PY
import collections, csv, json, re
calls = 11.0   # tools/bench_configs.py: 1 + 10 forward calls, 1 + 10 backward calls
acc = {"FETCH_SIZE": collections.defaultdict(list), "WRITE_SIZE": collections.defaultdict(list)}
for ctr in acc:
    try:
        rows = csv.DictReader(open("$D/steps_${c}_%s/p_counter_collection.csv" % ctr))
    except OSError:
        continue
    for r in rows:
        if r["Counter_Name"] != ctr or "mdconv::" not in r["Kernel_Name"].split("(")[0]:
            continue
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("mdconv::(anonymous namespace)::", "").replace("mdconv::", ""))
        acc[ctr][k].append(float(r["Counter_Value"]) * 1024)   # KiB -> bytes
kern = {}
for k in set(acc["FETCH_SIZE"]) | set(acc["WRITE_SIZE"]):
    f, w = acc["FETCH_SIZE"].get(k, []), acc["WRITE_SIZE"].get(k, [])
    kern[k] = {"launches_per_step": max(len(f), len(w)) / calls, "fetch_bytes_raw": sum(f) / calls, "write_bytes": sum(w) / calls}
out = {"fetch_bytes_raw": sum(v["fetch_bytes_raw"] for v in kern.values()), "write_bytes": sum(v["write_bytes"] for v in kern.values()),
       "kernels": {k: {kk: round(vv, 1) for kk, vv in v.items()} for k, v in sorted(kern.items(), key=lambda kv: -(2 * kv[1]["fetch_bytes_raw"] + kv[1]["write_bytes"]))[:8]}}
json.dump(out, open("$D/steps_$c.json", "w"), indent=1)
print("$c: fetch raw %.1f MB, write %.1f MB per step" % (out["fetch_bytes_raw"] / 1e6, out["write_bytes"] / 1e6))
PY
  rm -rf $D/steps_${c}_FETCH_SIZE $D/steps_${c}_WRITE_SIZE
done
