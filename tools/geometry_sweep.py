#!/usr/bin/env python3
"""Developer aid: one layer width over kernel extents / strides / dilations -- ms per step and ns per (output pixel x tap),
to spot geometries that fall off a fast path.   usage: python tools/geometry_sweep.py [2d|3d] [f32|f16] [C]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.cases import D2, D3, M2, M3, _c, out_size
from tools.anomaly_sweep import time_case

nd = 3 if "3d" in sys.argv else 2
dtype = torch.float16 if "f16" in sys.argv else torch.float32
C = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 64
sz, B = ((8, 24, 24), 2) if nd == 3 else ((48, 48), 8)
for op in ((M3, D3) if nd == 3 else (M2, D2)):
    for k in ((1, 2, 3) if nd == 3 else (1, 2, 3, 5, 7)):
        for stride in (1, 2):
            for dil in (1, 2):
                if k == 1 and dil == 2:
                    continue
                pad = dil * (k - 1) // 2
                case = _c("geo", op, B, C, C, sz, k, stride=stride, padding=pad, dilation=dil, tier="medium", seed=1)
                ms, p = time_case(case, dtype)
                n = B
                for v in out_size(case):
                    n *= v
                K = k ** nd
                print("%dd %s %-22s C=%d k=%d stride=%d dil=%d  N=%6d taps=%3d  %7.3f ms  %7.2f ns/(pixel tap)%s" % (
                    nd, str(dtype)[6:], op, C, k, stride, dil, n, K, ms, ms * 1e6 / (n * K),
                    "" if p == ["mfma", "mfma"] else "  " + str(p)), flush=True)
