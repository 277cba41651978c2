#!/usr/bin/env python3
"""Developer aid: which random shapes of tools/fuzz_more.py leave the matrix-core kernels (per direction), grouped by the
property that sends them to the shape-generic path.   usage: python tools/path_census.py [first] [count]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.cases import make_inputs
from tests.util import run_product
import tools.fuzz_more as F
first = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rows = collections.Counter()
examples = {}
for seed in range(first, first + count):
    case = F.case_f32(seed)
    t = make_inputs(case, device="cuda")
    _, _, p = run_product(case, t, "auto")
    C, O, G, DG = case["C"], case["O"], case["groups"], case["dgroups"]
    key = (p[0], p[1], "Cg=%d" % (C // G) if C // G < 16 else "Cg>=16", "Og=%d" % (O // G) if O // G < 16 else "Og>=16",
           "Cdg=%d" % (C // DG) if DG > 1 else "DG=1", "C%%8=%d" % (C % 8), "nd=%d" % len(case["in_sz"]))
    rows[key] += 1
    examples.setdefault(key, (C, O, G, DG, case["in_sz"], case["k"]))
for key, n in sorted(rows.items(), key=lambda kv: -kv[1]):
    if key[0] != "mfma" or key[1] != "mfma":
        print(n, key, "e.g.", examples[key])
print("all-matrix:", sum(n for k, n in rows.items() if k[0] == "mfma" and k[1] == "mfma"), "of", count)
