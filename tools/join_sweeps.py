#!/usr/bin/env python3
"""Developer aid: join two outputs of tools/realistic_sweep.py (same shape list, two trees run on ONE box) into a table.
usage: python tools/join_sweeps.py old.txt new.txt [old-label new-label]"""
import re
import sys


def rows(path):
    out = []
    for line in open(path):
        m = re.match(r"(\S+)\s+(\S+)\s+B=(\d+)\s+C=(\d+)\s+O=(\d+)\s+(\S+)\s+DG=(\d+)\s+([\d.]+) ms", line)
        if m:
            out.append((m.groups()[:7], float(m.group(8)), line.split("<--")[1].strip() if "<--" in line else ""))
    return out


def main():
    a, b = rows(sys.argv[1]), rows(sys.argv[2])
    la, lb = (sys.argv[3], sys.argv[4]) if len(sys.argv) > 4 else ("old", "new")
    assert [r[0] for r in a] == [r[0] for r in b]
    one = {}
    print("%-7s %-9s %-3s %-5s %-5s %-14s %-3s %9s %9s %7s %8s" % ("op", "dtype", "B", "C_in", "C_out", "spatial", "DG", la, lb, "ratio", "DG/DG=1"))
    for (k, ta, _), (_, tb, flag) in zip(a, b):
        op, dt, B, C, O, sz, dg = k
        if dg == "1":
            one[k[:6]] = tb
        rel = "%.2f" % (tb / one[k[:6]]) if dg != "1" and k[:6] in one else ""
        print("%-7s %-9s %-3s %-5s %-5s %-14s %-3s %9.3f %9.3f %7.2f %8s%s" % (op, dt, B, C, O, sz, dg, ta, tb, tb / ta, rel,
                                                                             ("  <-- " + flag) if flag else ""))


if __name__ == "__main__":
    main()
