#!/usr/bin/env python3
"""Developer aid: run ONE layer shape forward + backward a few times (for `rocprofv3 --kernel-trace --stats`) and print
its median step time.

    python tools/prof_shape.py m2:f16:B8:C256:O256:56x56:dg1[:g1][:k3][:d1] [more specs ...] [--n 10]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tests.cases import D2, D3, M2, M3, _c, make_inputs  # noqa: E402
from tests.util import run_product  # noqa: E402

OPS = {"d2": D2, "m2": M2, "d3": D3, "m3": M3}
DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


def parse(spec):
    f = spec.split(":")
    op, dtype = OPS[f[0]], DT[f[1]]
    kw = dict(B=1, C=64, O=64, sz=(16, 16), dg=1, g=1, k=3, d=1)
    for x in f[2:]:
        if x[0] == "B":
            kw["B"] = int(x[1:])
        elif x[0] == "C":
            kw["C"] = int(x[1:])
        elif x[0] == "O":
            kw["O"] = int(x[1:])
        elif x.startswith("dg"):
            kw["dg"] = int(x[2:])
        elif x[0] == "g":
            kw["g"] = int(x[1:])
        elif x[0] == "k":
            kw["k"] = int(x[1:])
        elif x[0] == "d":
            kw["d"] = int(x[1:])
        else:
            kw["sz"] = tuple(int(v) for v in x.split("x"))
    case = _c(spec, op, kw["B"], kw["C"], kw["O"], kw["sz"], kw["k"], padding=kw["d"] * (kw["k"] // 2), dilation=kw["d"],
              groups=kw["g"], dgroups=kw["dg"], tier="medium", seed=1)
    return case, dtype


def main():
    n = 10
    specs = []
    args = sys.argv[1:]
    i = 0
    while i < len(args):
        if args[i] == "--n":
            n = int(args[i + 1]); i += 2
        else:
            specs.append(args[i]); i += 1
    for spec in specs:
        case, dtype = parse(spec)
        t = make_inputs(case, dtype=dtype, device="cuda")
        for _ in range(2):
            _, _, p = run_product(case, t, "auto")
        torch.cuda.synchronize()
        times = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run_product(case, t, "auto")
            e1.record(); torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
        print("%-48s %8.3f ms  paths %s" % (spec, sorted(times)[len(times) // 2], p), flush=True)
        del t
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
