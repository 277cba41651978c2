#!/usr/bin/env python3
"""Developer aid: find layer shapes that run slower than a LARGER, friendlier shape does (channel counts rounded up to the
kernels' natural sizes): such a shape has fallen off a fast path and would be better off padded.
usage: python tools/anomaly_sweep.py [2d|3d] [f32|f16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.cases import D2, D3, M2, M3, _c, make_inputs
from tests.util import run_product

DT = {"f32": torch.float32, "f16": torch.float16}


def time_case(case, dtype, n=9):
    t = make_inputs(case, dtype=dtype, device="cuda")
    for _ in range(2):
        _, _, p = run_product(case, t, "auto")
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run_product(case, t, "auto"); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2], p


def main():
    nd = 3 if "3d" in sys.argv else 2
    dtype = DT["f16" if "f16" in sys.argv else "f32"]
    op = M3 if nd == 3 else M2
    sz, B = ((8, 20, 20), 2) if nd == 3 else ((40, 40), 8)
    if "small" in sys.argv:
        sz, B = ((4, 7, 7), 4) if nd == 3 else ((14, 14), 8)
    if "large" in sys.argv:
        sz, B = ((8, 40, 40), 2) if nd == 3 else ((112, 112), 4)
    chans = [3, 8, 16, 24, 40, 64, 72, 100, 128, 136, 200, 256, 264, 320, 512]
    outs = [4, 16, 40, 64, 100, 256]
    if "large" in sys.argv:
        chans, outs = [8, 16, 24, 40, 64, 72, 100, 128, 200, 256], [4, 16, 64, 100, 256]
    cache = {}

    def t_of(C, O, G, DG):
        key = (C, O, G, DG)
        if key not in cache:
            case = _c("an", op, B, C, O, sz, 3, groups=G, dgroups=DG, tier="medium", seed=1)
            cache[key] = time_case(case, dtype)
        return cache[key]

    def up(v, m):
        return (v + m - 1) // m * m

    combos = ((1, 1), (1, 2), (1, 4), (2, 1), (4, 1), (4, 4), (8, 2))
    if "g1" in sys.argv:
        combos = combos[:3]
    if "groups" in sys.argv:
        combos = combos[3:]
    for G, DG in combos:
        for C in chans:
            for O in outs:
                if C % G or O % G or C % DG:
                    continue
                ms, p = t_of(C, O, G, DG)
                # friendlier neighbour: channels per group up to 64s (per deformable group too), outputs per group up to 16s
                m = 64 * G * DG // __import__("math").gcd(G, DG)
                Cn, On = up(C, m), up(O, 16 * G)
                note = ""
                if (Cn, On) != (C, O):
                    msn, _ = t_of(Cn, On, G, DG)
                    if ms > 1.25 * msn:
                        note = "   <<< %.2fx its friendlier neighbour %d -> %d (%.3f ms)" % (ms / msn, Cn, On, msn)
                flag = "" if p == ["mfma", "mfma"] else "  " + str(p)
                print("%dd %s G=%d DG=%d  %4d -> %4d   %7.3f ms%s%s" % (nd, sys.argv[-1] if len(sys.argv) > 1 else "", G, DG, C, O, ms, flag, note), flush=True)


if __name__ == "__main__":
    main()
