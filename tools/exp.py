#!/usr/bin/env python3
"""Developer aid: one line per run -- step / forward / backward time of a bench.py workload (HIP events over N eager
steps) plus the in-library per-kernel events.  Environment knobs are read once per process, so every A/B leg is its
own process:   MDCONV_BWD_FORK=0 python tools/exp.py cfg2 [cfg2:4 ...] [--steps 20] [--label text]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from modulated_deform_conv_amd import _capi  # noqa: E402


def main():
    args = sys.argv[1:]
    steps, label, names = 20, "", []
    i = 0
    while i < len(args):
        if args[i] == "--steps":
            steps = int(args[i + 1]); i += 2
        elif args[i] == "--label":
            label = args[i + 1]; i += 2
        else:
            names.append(args[i]); i += 1
    for name in names or ["cfg2"]:
        base, _, b = name.partition(":")
        wl = bench.Workload(base, "cuda", int(b) if b else None)
        for _ in range(3):
            wl.forward(); wl.backward()
        torch.cuda.synchronize()
        _capi.profile_enable(True)
        _capi.profile_reset()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        for _ in range(steps):
            wl.forward(); wl.backward()
        ev[1].record()
        for _ in range(steps):
            wl.forward()
        ev[2].record()
        for _ in range(steps):
            wl.backward()
        ev[3].record()
        torch.cuda.synchronize()
        _capi.profile_enable(False)
        prof = _capi.profile_read()
        t = [ev[k].elapsed_time(ev[k + 1]) / steps for k in range(3)]
        print("%-28s %-8s step %.3f ms  fwd %.3f  bwd %.3f  | %s" % (
            label, name, t[0], t[1], t[2], "  ".join("%s %.3f" % (k.replace("_kernel", ""), v[1]) for k, v in prof.items())),
            flush=True)
        del wl
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
