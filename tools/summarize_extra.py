#!/usr/bin/env python3
"""Turn gpurun_out/<tag>_extra/ (tools/collect_extra.sh, run on the GPU box) into the two committed
summaries profiles/<tag>_other_configs.md and profiles/<tag>_hp_counters.md.
usage: python tools/summarize_extra.py r03"""
import collections
import csv
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", tag + "_extra")
prof = os.path.join(ROOT, "profiles")


def read(name):
    p = os.path.join(src, name)
    return open(p).read() if os.path.exists(p) else ""


def counters(name):
    """one line per kernel: 'kernel A=1 B=2' -> {kernel: {A: 1.0, B: 2.0}}"""
    out = collections.OrderedDict()
    for line in read(name).splitlines():
        m = re.match(r"^(\S.*?) ((?:[A-Za-z0-9_]+=[-+.e0-9]+ ?)+)$", line.strip())
        if m:
            out[m.group(1)] = {k: float(v) for k, v in (kv.split("=") for kv in m.group(2).split())}
    return out


def stats(cfg):
    keep = [l for l in read("stats_%s.txt" % cfg).splitlines() if l.startswith("==") or " calls " in l]
    return "\n".join(keep)


# ---------------------------------------------------------------- other configs
o = ["# %s -- the other BASELINE.json configurations on one MI355X (not bench lines)" % tag, "",
     "Collected by `tools/collect_extra.sh` on the GPU box (same build as `profiles/%s_summary.md`); per-GPU" % tag,
     "shards of the configurations, default kernel path, mean of 10 calls after a warm-up",
     "(`tools/bench_configs.py`, the workload definitions of `bench.py`).  The eager lines go through the same",
     "allocating Python entry points as `bench.py`; short eager loops of small shards expose the host's launch",
     "latency (about 14 launches and 7 allocations per step), so the strong-scaling shards are quoted from the",
     "`--graph` lines (a captured step replayed 20 times).", "",
     "```", read("configs.txt").strip(), "```", "",
     "The kernel-path tag printed by the script is the library's `mdconv_last_path`: `mfma` covers both the",
     "fp32 MFMA kernels and the native 16-bit (`hp_*`) kernels; the kernel names below tell them apart.", "",
     "Kernel breakdown (`tools/prof_cfg.sh`: `rocprofv3 --kernel-trace --stats` of the same script; the",
     "`float16_copy` / `FillFunctor` elementwise kernels are the script's own input preparation, outside",
     "the timed calls):", "", "```"]
for c in ("cfg3", "cfg4", "cfg5"):
    o.append(stats(c))
o += ["```", ""]


def traffic_table(cfg, pref_note):
    f, w = counters("fetch_%s.txt" % cfg), counters("write_%s.txt" % cfg)
    if not f:
        return []
    t = ["HBM-side traffic per launch at %s (separate `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` passes, KiB" % cfg,
         "counters -> MB; read side raw and x2, see the calibration note in `%s_hp_counters.md`)%s:" % (tag, pref_note), "",
         "| kernel | FETCH raw MB | FETCH x2 MB | WRITE MB |", "|---|---|---|---|"]
    rows = []
    for k, v in f.items():
        fr = v.get("FETCH_SIZE", 0.0) * 1024 / 1e6
        wr = w.get(k, {}).get("WRITE_SIZE", 0.0) * 1024 / 1e6
        rows.append((fr + wr, "| %s | %.1f | %.1f | %.1f |" % (k, fr, 2 * fr, wr)))
    t += [r for _, r in sorted(rows, reverse=True) if _ > 1.0]
    return t + [""]


for c in ("cfg3", "cfg5", "cfg4"):
    o += traffic_table(c, "")
open(os.path.join(prof, tag + "_other_configs.md"), "w").write("\n".join(o) + "\n")

# ---------------------------------------------------------------- counters
h = ["# %s -- SQ / GRBM counters of the 16-bit and 3-D kernels, FETCH_SIZE calibration, microbenchmarks" % tag, "",
     "All from separate `rocprofv3 --pmc` passes (`tools/pmc_cfg.sh`, no tracing in the same run) over",
     "`tools/bench_configs.py <cfg>`; sums over the chip, averaged per launch.", "",
     "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); VALU = SQ_INSTS_VALU -",
     "SQ_INSTS_MFMA; waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES.", ""]
for c in ("cfg3", "cfg5", "cfg4"):
    sq, gr = counters("sq_%s.txt" % c), counters("grbm_%s.txt" % c)
    if not sq:
        continue
    h += ["## %s" % c, "",
          "| kernel | MFMA instr | other VALU instr | VALU per SIMD | kernel cycles (per XCD) | MfmaUtil | waves waiting | TA busy |",
          "|---|---|---|---|---|---|---|---|"]
    for k, v in sq.items():
        g = gr.get(k, {})
        cyc = g.get("GRBM_GUI_ACTIVE", 0.0) / 8
        if cyc < 2e4:
            continue
        valu = v.get("INSTS_VALU", 0.0) - v.get("INSTS_MFMA", 0.0)
        util = v.get("VALU_MFMA_BUSY_C", 0.0) / (1024 * cyc) if cyc else 0.0
        wait = v.get("WAIT_ANY", 0.0) / v["WAVE_C"] if v.get("WAVE_C") else 0.0
        ta = g.get("GRBM_TA_BUSY", 0.0) / g["GRBM_GUI_ACTIVE"] if g.get("GRBM_GUI_ACTIVE") else 0.0
        h.append("| %s | %.3g | %.3g | %.3g | %.3g | %.0f %% | %.0f %% | %.0f %% |" %
                 (k, v.get("INSTS_MFMA", 0.0), valu, valu / 1024, cyc, 100 * util, 100 * wait, 100 * ta))
    h.append("")

h += ["## Cache path (TCP = vector L1, TCC = L2; 128-byte L2 requests)", "",
      "| config | kernel | vector loads (wave instr) | L1 accesses | L1 accesses per load | L1 -> L2 reads | L2 requests | L2 hit |",
      "|---|---|---|---|---|---|---|---|"]
for c in ("cfg3", "cfg4", "cfg5"):
    tcp, tcc, vm = counters("tcp_%s.txt" % c), counters("tcc_%s.txt" % c), counters("vmem_%s.txt" % c)
    for k, v in tcp.items():
        acc = v.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0)
        if acc < 2e7:
            continue
        rd = vm.get(k, {}).get("INSTS_VMEM_RD", 0.0)
        t = tcc.get(k, {})
        hit = t.get("TCC_HIT_sum", 0.0) / t["TCC_REQ_sum"] if t.get("TCC_REQ_sum") else 0.0
        h.append("| %s | %s | %.3g | %.3g | %.1f | %.3g | %.3g | %.0f %% |" %
                 (c, k, rd, acc, acc / rd if rd else 0.0, v.get("TCP_TCC_READ_REQ_sum", 0.0), t.get("TCC_REQ_sum", 0.0), 100 * hit))
h.append("")

h += ["## FETCH_SIZE calibration (`tools/ubench_fetch.hip`: every pattern reads the same 1 GiB buffer exactly once)", "", "```"]
p = os.path.join(src, "fetchcal_counter_collection.csv")
if os.path.exists(p):
    for r in csv.DictReader(open(p)):
        if r["Counter_Name"] == "FETCH_SIZE":
            name = re.sub(r"\(.*", "", r["Kernel_Name"])
            h.append("%-28s FETCH_SIZE = %10.0f KiB = %.3f of the 1 GiB read" % (name, float(r["Counter_Value"]), float(r["Counter_Value"]) / (1 << 20)))
h += ["```", "",
      "Coalesced 4- / 8- / 16-byte-per-lane streams report exactly 1/2 (128-byte requests tallied at 64 B);",
      "line-granular 8-byte gathers report their real request size.  The x2 correction of the microarch",
      "guide therefore holds for streams and over-states gather-dominated kernels by up to 2x; the",
      "summaries quote raw and x2 side by side.", "",
      "## Texture-path microbenchmark (`tools/ubench_gather16.hip`, cycles per 64-lane 16-byte load per CU)", "", "```",
      read("ubench_gather16.txt").strip(), "```", ""]
open(os.path.join(prof, tag + "_hp_counters.md"), "w").write("\n".join(h) + "\n")
print("wrote", tag + "_other_configs.md", tag + "_hp_counters.md")
