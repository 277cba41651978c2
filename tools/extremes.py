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
from tests.cases import EXTREME_F32 as F32, EXTREME_HP as HP, make_inputs  # noqa: E402
from tests.util import elem_err, rel_err, run_oracle, run_product  # noqa: E402

spec = importlib.util.spec_from_file_location("fuzz_more", os.path.join(ROOT, "tools", "fuzz_more.py"))
fm = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fm)

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
