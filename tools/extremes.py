#!/usr/bin/env python3
"""Shapes at the edges of the supported range (developer aid, GPU box): very many taps, very long axes, very wide channel
counts, very many images of one pixel.  Each runs through the default kernel selection and through the shape-generic kernels
(two independent implementations, fp32; for the 16-bit shapes: native kernels against the fp32 kernels' result on the same
rounded inputs) with NaN margins around every input and guard margins around caller-allocated outputs (tools/fuzz_more.py),
and the small ones also against the CPU oracle.  Prints one line per shape; exit status 1 if any fails.
usage: python tools/extremes.py"""
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from tests.cases import D2, D3, M2, M3, _c, make_inputs  # noqa: E402
from tests.util import elem_err, rel_err, run_oracle, run_product  # noqa: E402

spec = importlib.util.spec_from_file_location("fuzz_more", os.path.join(ROOT, "tools", "fuzz_more.py"))
fm = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fm)

F32 = [
    # many taps
    _c("x_k11_2d", M2, 2, 16, 16, (20, 24), 11, padding=5, seed=1),
    _c("x_k31_2d", D2, 1, 16, 16, (36, 33), 31, padding=0, seed=2),
    _c("x_k35_2d_1225taps", M2, 1, 16, 16, (36, 38), 35, padding=0, seed=3, bias=False),   # generic backward: <= 1280 taps
    _c("x_k9_3d_729taps", M3, 1, 16, 16, (9, 10, 11), 9, padding=0, seed=4),
    _c("x_k7_3d_343taps_c64", D3, 2, 64, 64, (8, 8, 8), 7, padding=3, seed=5),
    _c("x_k1x49_2d", M2, 2, 32, 32, (5, 60), (1, 49), padding=(0, 0), seed=6),
    # long axes / many pixels
    _c("x_2d_1x65536", M2, 1, 16, 16, (1, 65536), (1, 3), padding=(0, 1), seed=7),
    _c("x_2d_3000x3000_c16", D2, 1, 16, 16, (3000, 3000), 3, seed=8, bias=False),
    _c("x_3d_256x4x4", M3, 1, 32, 32, (256, 4, 4), 3, seed=9),
    _c("x_2d_stride7_dil9", M2, 2, 32, 32, (80, 80), 3, stride=7, dilation=9, padding=9, seed=10),
    # wide channel counts
    _c("x_c2048_o2048_8x8", M2, 2, 2048, 2048, (8, 8), 3, seed=11),
    _c("x_c4096_o16_6x6_dg64", M2, 1, 4096, 16, (6, 6), 3, dgroups=64, seed=12),
    _c("x_c16_o4096_6x6", D2, 1, 16, 4096, (6, 6), 3, seed=13),
    _c("x_c1024_g32_dg32_3d", M3, 1, 1024, 1024, (4, 4, 4), 3, groups=32, dgroups=32, seed=14),
    # many images of one output position
    _c("x_b4096_1x1out", M2, 4096, 32, 32, (3, 3), 3, padding=0, seed=15),
    _c("x_b70000_1x1_k1", D2, 70000, 16, 16, (1, 1), 1, padding=0, seed=16),
    _c("x_b513_3d_1out", M3, 513, 32, 48, (3, 3, 3), 3, padding=0, seed=17),
    # offsets far outside the image
    _c("x_offsets_100px", M2, 2, 64, 64, (20, 20), 3, seed=18, offset_scale=100.0),
    _c("x_offsets_1e6px", D2, 2, 64, 64, (20, 20), 3, seed=19, offset_scale=1e6),
]
HP = [
    _c("xh_k7_2d", M2, 2, 64, 64, (20, 20), 7, padding=3, seed=31),
    _c("xh_k5_3d_125taps", M3, 1, 128, 128, (6, 8, 8), 5, padding=2, seed=32),
    _c("xh_k1x25", M2, 2, 128, 64, (4, 40), (1, 25), padding=0, seed=33),
    _c("xh_c2048_7x7", M2, 2, 2048, 512, (7, 7), 3, seed=34),
    _c("xh_c1024_dg16_g4", M2, 1, 1024, 256, (10, 10), 3, groups=4, dgroups=16, seed=35),
    _c("xh_2d_1x32768", M2, 1, 64, 64, (1, 32768), (1, 3), padding=(0, 1), seed=36),
    _c("xh_2d_1500x1500_c32", D2, 1, 32, 32, (1500, 1500), 3, seed=37),
    _c("xh_b4096_1x1out", M2, 4096, 64, 64, (3, 3), 3, padding=0, seed=38),
    _c("xh_3d_stride3_dil3", D3, 2, 64, 64, (20, 20, 20), 3, stride=3, dilation=3, padding=3, seed=39),
    _c("xh_offsets_100px", M3, 1, 64, 64, (8, 8, 8), 3, seed=40, offset_scale=100.0),
]


# fp32 coordinates beyond 2^15 pixels: the reference's (and the oracle's) `(p + 1 - high)` rounds once more where p + 1 crosses
# a power of two, the kernels use `p - low`: grad_input differs by up to an ulp of the coordinate at the columns next to 2^k
# and nowhere else (INTEGRATION.md "Limits"; profiles/r05_experiments.md 26)
ORACLE_TOL = {"x_2d_1x65536": 5e-3}


def errs(got, gots, want, wants):
    worst = (0.0, 0.0, "")
    for k, v in [("output", got)] + sorted(gots.items()):
        w = want if k == "output" else wants[k]
        if v is None or w is None:
            continue
        e, pe = rel_err(v.float(), w.float()), elem_err(v.float(), w.float())
        if pe > worst[1]:
            d = (v.float().cpu() - w.float().cpu()).abs().flatten()
            i = int(d.argmax())
            worst = (e, pe, "%s, largest difference %.2e at flat index %d = last-axis coordinate %d" % (k, d[i], i, i % v.shape[-1]))
    return worst


def main():
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    bad = 0
    for case in F32 + HP:
        if only and not any(o in case["name"] for o in only):
            continue
        hp = case in HP
        dtype = torch.float16 if hp else torch.float32
        t0 = time.time()
        try:
            t = fm.nan_margined(make_inputs(case, dtype=dtype, device="cuda"))
            out_a, g_a, p = run_product(case, t, "auto")
            torch.cuda.synchronize()
            nf = fm.all_finite(out_a, g_a)
            if hp:
                # second implementation: the fp32 kernels on the same (rounded) values
                t32 = {k: (None if v is None else v.float()) for k, v in t.items()}
                out_b, g_b, _ = run_product(case, t32, "auto")
                tol = 1e-2
            else:
                out_b, g_b, _ = run_product(case, t, "direct")
                tol = 1e-4
            torch.cuda.synchronize()
            e, pe, k = errs(out_a, g_a, out_b, g_b)
            viol = fm.run_guarded_outputs(case, t)
            msg = "%s vs second implementation: scaled %.1e per-element %.1e (%s)" % (p, e, pe, k)
            ok = not nf and not viol and e <= tol and pe <= tol
            work = case["B"] * case["C"] * case["O"] * torch.tensor(case["in_sz"]).prod().item()
            if work < 3e7 and not hp:
                want_out, want = run_oracle(case, t, torch.float32)
                eo, peo, ko = errs(out_a, g_a, want_out, want)
                msg += "; vs oracle: %.1e / %.1e (%s)" % (eo, peo, ko)
                otol = ORACLE_TOL.get(case["name"], tol)
                ok = ok and eo <= otol and peo <= otol
            if nf:
                msg += "; NON-FINITE " + str(nf)
            if viol:
                msg += "; GUARD " + str(viol)
        except Exception as ex:   # unsupported shapes must say so, not crash
            ok, msg = False, "%s: %s" % (type(ex).__name__, str(ex).split("\n")[0][:200])
        print("%s %-28s %s  [%.1f s]" % ("ok  " if ok else "FAIL", case["name"], msg, time.time() - t0), flush=True)
        bad += 0 if ok else 1
        del t
        torch.cuda.empty_cache()
    print("extremes: %d shapes, %d FAILED" % (len(F32 + HP) if not only else -1, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
