#!/usr/bin/env python3
"""Where one step's wall time goes, from a rocprofv3 --kernel-trace CSV: per kernel of the LAST complete step its
start offset, duration and the idle gap before it on the timeline of all queues merged (overlapping kernels of the
forked stream show a negative gap).   usage: tools/gap_report.py <..._kernel_trace.csv> [first-kernel-substring]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "pack_weights"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
              re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("mdconv::(anonymous namespace)::", "")),
              r.get("Queue_Id", "")) for r in rows), key=lambda e: e[0])
starts = [i for i, e in enumerate(ev) if first in e[2]]
if len(starts) < 3:
    sys.exit("fewer than 3 steps found")
a, b = starts[-2], starts[-1]
t0, busy_end = ev[a][0], ev[a][0]
print("step of %d kernels, %.3f ms from its first kernel to the next step's first kernel" % (b - a, (ev[b][0] - t0) / 1e6))
for s, e, name, q in ev[a:b]:
    print("  +%8.1f us  %8.1f us  gap %+7.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - busy_end) / 1e3, q, name[:70]))
    busy_end = max(busy_end, e)
print("  tail gap to the next step: %+.1f us" % ((ev[b][0] - busy_end) / 1e3))
