#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs of one bench.py profiling session (gpurun_out/prof_rNN/) into the
summaries committed under profiles/:  rNN_kernel_stats.csv (verbatim --stats table, names
shortened), rNN_pmc_summary.json (per-kernel FETCH_SIZE / WRITE_SIZE per launch, separate --pmc
passes) and rNN_summary.md.

HBM bytes follow MI355X_MICROARCH.md section HBM: counters are in KiB; on gfx950 FETCH_SIZE reports
half the bytes of a wide coalesced stream, so the read side is given both raw and doubled."""
import collections
import csv
import json
import os
import re
import sys


def short(name):
    name = name.replace("void ", "").replace("mdconv::(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", name)


def sq_table(src):
    """MfmaUtil etc. from the SQ / GRBM passes of tools/collect_profiles.sh (sq_counters.txt)."""
    import ast
    path = os.path.join(src, "sq_counters.txt")
    if not os.path.exists(path):
        return ""
    rows = collections.defaultdict(dict)
    for line in open(path):
        m = re.match(r"^(\S.*?) (\{.*\})\s*$", line)
        if m:
            rows[m.group(1)].update({k: float(v) for k, v in ast.literal_eval(m.group(2)).items()})
    out = ["", "## SQ / GRBM counters (separate `rocprofv3 --pmc` passes, `tools/pmc_kernel.sh`; sums over the chip, per launch)", "",
           "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); VALU = SQ_INSTS_VALU - SQ_INSTS_MFMA.", "",
           "| kernel | MFMA instr | other VALU instr | VALU per MFMA | kernel cycles (per XCD) | MfmaUtil | wave time waiting for issue |",
           "|---|---|---|---|---|---|---|"]
    for k, v in rows.items():
        if not v.get("SQ_INSTS_MFMA") or not v.get("GRBM_GUI_ACTIVE"):
            continue
        cyc = v["GRBM_GUI_ACTIVE"] / 8
        valu = v["SQ_INSTS_VALU"] - v["SQ_INSTS_MFMA"]
        out.append("| %s | %.3g | %.3g | %.2f | %.3g | %.0f %% | %.0f %% |" % (
            k, v["SQ_INSTS_MFMA"], valu, valu / v["SQ_INSTS_MFMA"], cyc,
            100 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc),
            100 * v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"]))
    return "\n".join(out) + "\n"


def main(src, tag):
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(out_dir, exist_ok=True)
    stats = list(csv.DictReader(open(os.path.join(src, "bench_kernel_stats.csv"))))
    with open(os.path.join(out_dir, tag + "_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"])
        for r in stats:
            w.writerow([short(r["Name"])[:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                        r["Percentage"], r["MinNs"], r["MaxNs"]])
    pmc = collections.defaultdict(dict)
    for fname, counter in (("pmc_fetch_counter_collection.csv", "FETCH_SIZE"),
                           ("pmc_write_counter_collection.csv", "WRITE_SIZE")):
        path = os.path.join(src, fname)
        if not os.path.exists(path):
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            pmc[k][counter + "_KiB_per_launch"] = sum(v) / len(v)
            pmc[k]["launches_" + counter] = len(v)
    summary = {}
    for k, v in pmc.items():
        if not k.startswith(("mfma_", "col2im", "pack_", "csr_", "tap_", "grad_bias", "reduce_", "nchw_", "zero_", "bwd_prep", "fwd_tail")):
            continue
        f_raw = v.get("FETCH_SIZE_KiB_per_launch", 0.0) * 1024
        wr = v.get("WRITE_SIZE_KiB_per_launch", 0.0) * 1024
        summary[k] = {"fetch_bytes_raw": f_raw, "fetch_bytes_x2_gfx950": 2 * f_raw, "write_bytes": wr,
                      "hbm_bytes_per_launch": 2 * f_raw + wr}
    # stamp: hash of the kernel sources the counters were collected on (bench.py refuses the traffic
    # figure when the sources it times differ); collect_profiles.sh records it on the GPU box
    stamp_file = os.path.join(src, "kernel_sources_sha16.txt")
    if os.path.exists(stamp_file):
        summary["_kernel_sources_sha16"] = open(stamp_file).read().strip()
    # per-configuration step totals (tools/collect_steps.sh): bench.py's `traffic` fields of the headline and of other_configs
    steps = {}
    for cfg in ("cfg2", "cfg3", "cfg4", "cfg5"):
        p = os.path.join(src, "steps_%s.json" % cfg)
        if os.path.exists(p):
            steps[cfg] = json.load(open(p))
    if steps:
        summary["_steps"] = steps
    json.dump(summary, open(os.path.join(out_dir, tag + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)
    bench_json = None
    bj = os.path.join(src, "bench_full.json")
    if os.path.exists(bj):
        bench_json = open(bj).read().strip()
        open(os.path.join(out_dir, tag + "_bench.json"), "w").write(bench_json + "\n")
    with open(os.path.join(out_dir, tag + "_summary.md"), "w") as f:
        f.write("# %s -- rocprofv3 summary of `python bench.py` (MI355X, cfg2: MDCN2d 3x3 C=256 56x56 B=32 fp32)\n\n" % tag)
        f.write("Commands (on the GPU box, `cd /tmp && export TMPDIR=/tmp`):\n\n"
                "    rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs\n"
                "    rocprofv3 --pmc FETCH_SIZE --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs\n"
                "    rocprofv3 --pmc WRITE_SIZE --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs\n\n")
        f.write("## Kernel time (--kernel-trace --stats)\n\n| kernel | calls | avg us | % |\n|---|---|---|---|\n")
        for r in stats[:16]:
            f.write("| %s | %s | %.1f | %s |\n" % (short(r["Name"])[:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
        f.write("\n## HBM traffic per launch (separate --pmc passes; KiB counters -> bytes; read side x2 per the gfx950 note)\n\n"
                "| kernel | FETCH raw MB | FETCH x2 MB | WRITE MB | total MB |\n|---|---|---|---|---|\n")
        for k, v in sorted(((k, v) for k, v in summary.items() if isinstance(v, dict) and not k.startswith("_")), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]):
            f.write("| %s | %.1f | %.1f | %.1f | %.1f |\n" % (k[:60], v["fetch_bytes_raw"] / 1e6, v["fetch_bytes_x2_gfx950"] / 1e6,
                                                          v["write_bytes"] / 1e6, v["hbm_bytes_per_launch"] / 1e6))
        if steps:
            f.write("\n## Counter traffic of one forward + backward step, all kernels (tools/collect_steps.sh over tools/bench_configs.py)\n\n"
                    "| config | FETCH raw MB | FETCH x2 MB | WRITE MB | total (x2) MB |\n|---|---|---|---|---|\n")
            for cfg, v in steps.items():
                f.write("| %s | %.1f | %.1f | %.1f | %.1f |\n" % (cfg, v["fetch_bytes_raw"] / 1e6, 2 * v["fetch_bytes_raw"] / 1e6,
                                                                 v["write_bytes"] / 1e6, (2 * v["fetch_bytes_raw"] + v["write_bytes"]) / 1e6))
        if bench_json:
            f.write("\n## bench.py line of the same build (un-profiled run)\n\n```\n%s\n```\n" % bench_json)
        for extra, title in (("bench_graph.json", "same build, `--graph` (step replayed from a HIP graph)"),
                             ("bench_nofork.json", "same build, `MDCONV_BWD_FORK=0` (grad_input gather AFTER GEMM-2 on the caller's stream, the round-2 order)")):
            pe = os.path.join(src, extra)
            if os.path.exists(pe) and open(pe).read().strip():
                f.write("\n## %s\n\n```\n%s\n```\n" % (title, open(pe).read().strip()))
    print("wrote profiles/%s_*" % tag)


def append_sq(src, tag):
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    with open(os.path.join(out_dir, tag + "_summary.md"), "a") as f:
        f.write(sq_table(src))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
    append_sq(sys.argv[1], sys.argv[2])
