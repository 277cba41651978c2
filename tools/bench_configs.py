#!/usr/bin/env python3
"""Secondary timings (not the headline metric): forward / backward of the BASELINE.json
configurations on one GPU, default kernel path, same workload definitions as bench.py.

    python tools/bench_configs.py [cfg2|cfg3|cfg4|cfg5|cfg2:B] ...     (cfg2:16 = a 16-image shard of cfg2)

Eager timings of short loops include the host's launch latency (about 14 launches and 7 allocations per
step from Python); `--graph` replays a captured step instead -- the number to quote for small shards."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from modulated_deform_conv_amd import _capi  # noqa: E402


def timeit(fn, n):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def run(name, graph):
    base, _, b = name.partition(":")
    wl = bench.Workload(base, "cuda", int(b) if b else None)
    if graph:
        def both():
            wl.forward(); wl.backward()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            both(); both()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            both()
        t = timeit(g.replay, 20)
        print("%s: fwd+bwd %.3f ms (graph replay, %s) -> %.2f GSamples/s" % (name, t, _capi.last_kernels(), wl.n_samples / t / 1e6))
        return
    tf = timeit(wl.forward, 10); pf = _capi.last_kernels()
    tb = timeit(wl.backward, 10); pb = _capi.last_kernels()
    print("%s: fwd %.3f ms (%s)  bwd %.3f ms (%s)  -> %.2f GSamples/s" % (name, tf, pf, tb, pb, wl.n_samples / (tf + tb) / 1e6))


if __name__ == "__main__":
    names = [a for a in sys.argv[1:] if not a.startswith("--")]
    for n in (names or ["cfg4"]):
        run(n, "--graph" in sys.argv)
