#!/usr/bin/env python3
"""Direct measurement of `north_star`'s "per-in_step sub-batches on HIP streams" (VERDICT r4, row g1): the cfg2 step
(B = 32, forward + backward) against the SAME work cut into two half batches (B = 16 each) that go through the
unchanged C ABI on two caller streams of different priority, so that the forward of one half can run beside the
backward of the other and GEMM-1 of one half beside gather + GEMM-2 of the other.  Reference chunk loop:
src/config.h:43-60, mdeformable_conv.cu:167-182 (one chunk after the other on the null stream).

Legs (same process, same box, HIP events over `--steps` steps after a warm-up; `--graph` replays a captured step):
  whole      B = 32, one stream                                          (the shipped step)
  serial     two halves one after the other on one stream                (what chunking alone costs)
  parallel   half 0 on stream A, half 1 on stream B, both started together
  staggered  stream B starts its forward when stream A's forward is done: fwd(1) runs beside bwd(0), bwd(1) beside the
             tail of bwd(0); every step ends with a join of both streams, as a pipeline inside one call would
grad_weight / grad_bias of the halves are added (one small kernel) so that every leg produces the same results.

    python tools/subbatch_pipeline.py [--steps 20] [--graph] [--halves 2|4]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 20
    graph = "--graph" in sys.argv
    nh = int(sys.argv[sys.argv.index("--halves") + 1]) if "--halves" in sys.argv else 2
    name = "cfg2"
    whole = bench.Workload(name, "cuda")
    B = whole.B
    parts = []
    for i in range(nh):
        p = bench.Workload(name, "cuda")
        p.shard(i * B // nh, (i + 1) * B // nh)
        parts.append(p)
    streams = [torch.cuda.Stream(priority=(-1 if i % 2 == 0 else 0)) for i in range(nh)]

    def step_whole():
        whole.forward()
        return whole.backward()

    def step_serial():
        gw = gb = None
        for p in parts:
            p.forward()
            w, b = p.backward()
            gw = w if gw is None else gw + w
            gb = b if gb is None else gb + b
        return gw, gb

    def step_streams(stagger):
        res = []
        evs = []
        main_s = torch.cuda.current_stream()   # (the capture stream while a graph is being recorded)
        for i, (p, s) in enumerate(zip(parts, streams)):
            s.wait_stream(main_s)
            if stagger and i > 0:
                s.wait_event(evs[i - 1])
            with torch.cuda.stream(s):
                p.forward()
                if stagger:
                    e = torch.cuda.Event()
                    e.record(s)
                    evs.append(e)
                res.append(p.backward())
        for s in streams:
            main_s.wait_stream(s)
        for w, b in res:
            w.record_stream(main_s)
            b.record_stream(main_s)
        gw, gb = res[0]
        for w, b in res[1:]:
            gw, gb = gw + w, gb + b
        return gw, gb

    legs = [("whole", step_whole), ("serial", step_serial), ("parallel", lambda: step_streams(False)),
            ("staggered", lambda: step_streams(True))]
    # results agree (fp32 re-association of the batch sum only)
    ref = step_whole()
    torch.cuda.synchronize()
    for lname, fn in legs[1:]:
        got = fn()
        torch.cuda.synchronize()
        err = ((got[0] - ref[0]).abs().max() / ref[0].abs().max()).item()
        assert err < 1e-4, (lname, err)
    for lname, fn in legs:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        run = fn
        if graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            run = g.replay
            run()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            run()
        e1.record()
        torch.cuda.synchronize()
        print("%-10s %s  %d x B=%d  %.3f ms per step" % (lname, "graph" if graph else "eager", 1 if lname == "whole" else nh,
                                                       B if lname == "whole" else B // nh, e0.elapsed_time(e1) / steps), flush=True)


if __name__ == "__main__":
    main()
