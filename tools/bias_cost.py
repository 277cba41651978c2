#!/usr/bin/env python3
"""Developer aid: what `with_bias` costs per step -- the BASELINE shards (bench.py WORKLOADS; cfg3 / cfg4 / cfg5 are quoted without
bias, the reference's module default) timed with bias off and on.   usage: python tools/bias_cost.py [cfg3 cfg4 cfg5 cfg2]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402


def timed(name, bias, steps=10):
    bench.WORKLOADS[name]["bias"] = bias
    wl = bench.Workload(name, "cuda")
    for _ in range(3):
        wl.forward(); wl.backward()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    for _ in range(steps):
        wl.forward()
    ev[1].record()
    for _ in range(steps):
        wl.backward()
    ev[2].record()
    torch.cuda.synchronize()
    del wl
    torch.cuda.empty_cache()
    return ev[0].elapsed_time(ev[1]) / steps, ev[1].elapsed_time(ev[2]) / steps


for name in sys.argv[1:] or ["cfg3", "cfg4", "cfg5", "cfg2"]:
    for rep in range(2):
        f0, b0 = timed(name, False)
        f1, b1 = timed(name, True)
        print("%s  no bias: fwd %.3f bwd %.3f ms   bias: fwd %.3f bwd %.3f ms   (+%.3f / +%.3f ms)" % (name, f0, b0, f1, b1, f1 - f0, b1 - b0),
              flush=True)
