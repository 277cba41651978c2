#!/usr/bin/env python3
"""Secondary timings (not the headline metric): forward / backward of the other BASELINE.json
configurations on one GPU, default kernel path.  python tools/bench_configs.py [cfg4|cfg3|cfg5|cfg2]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modulated_deform_conv_amd import MDCONV_CUDA as M, _capi  # noqa: E402


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def run(name):
    g = torch.Generator().manual_seed(0)
    rn = lambda *s: torch.randn(*s, generator=g)
    if name == "cfg4":   # DeformConv3d 3x3x3 C=64 32^3 B=8 fp32
        B, C, O, sp, K, dt = 8, 64, 64, (32, 32, 32), 27, torch.float32
        x, off = rn(B, C, *sp), rn(B, 3 * K, *sp)
        w = (torch.rand(O, C, 3, 3, 3, generator=g) * 2 - 1) / math.sqrt(C * K)
        go = rn(B, O, *sp)
        x, off, w, go = [t.cuda().to(dt).contiguous() for t in (x, off, w, go)]
        b = x.new_empty(0)
        geo = (3, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 64, False)
        out = torch.empty_like(go)
        f = lambda: M.deform_conv3d_forward_cuda(x, w, b, off, out, *geo)
        gi, gw, gb, goff = torch.zeros_like(x), torch.zeros_like(w), torch.zeros_like(b), torch.zeros_like(off)
        bw = lambda: M.deform_conv3d_backward_cuda(x, w, b, off, gi, gw, gb, goff, go, *geo)
        ns = B * C * K * math.prod(sp)
    elif name.startswith("cfg2"):   # "cfg2" or "cfg2:B" (strong-scaling shards: cfg2:16, cfg2:8, cfg2:4)
        B, C, O, sp, K = 32, 256, 256, (56, 56), 9
        if ":" in name:
            B = int(name.split(":")[1])
        x, off, m = rn(B, C, *sp).cuda(), rn(B, 2 * K, *sp).cuda(), torch.sigmoid(rn(B, K, *sp)).cuda()
        w = ((torch.rand(O, C, 3, 3, generator=g) * 2 - 1) / math.sqrt(C * K)).cuda()
        b, go = (0.1 * rn(O)).cuda(), rn(B, O, *sp).cuda()
        geo = (3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 64, True)
        f = lambda: M.modulated_deform_conv2d_forward_cuda(x, w, b, off, m, *geo)
        bw = lambda: M.modulated_deform_conv2d_backward_cuda(x, w, b, off, m, go, *geo)
        ns = B * C * K * math.prod(sp)
    elif name == "cfg3":   # MDCN2d C=256 56x56 B=32/GPU G=32 DG=4 fp16
        B, C, O, G, DG, K, sp = 32, 256, 256, 32, 4, 9, (56, 56)
        h = lambda t: t.cuda().half().contiguous()
        x, off, m = h(rn(B, C, *sp)), h(rn(B, DG * 2 * K, *sp)), h(torch.sigmoid(rn(B, DG * K, *sp)))
        w = h((torch.rand(O, C // G, 3, 3, generator=g) * 2 - 1) / math.sqrt(C * K))
        b, go = x.new_empty(0), h(rn(B, O, *sp))
        geo = (3, 3, 1, 1, 1, 1, 1, 1, G, DG, 64, False)
        f = lambda: M.modulated_deform_conv2d_forward_cuda(x, w, b, off, m, *geo)
        bw = lambda: M.modulated_deform_conv2d_backward_cuda(x, w, b, off, m, go, *geo)
        ns = B * C * K * math.prod(sp)
    elif name in ("cfg5", "cfg5z", "cfg5l2"):   # MDCN3d C=128 16x64x64 B=8/GPU dil 2 fp16
        # cfg5z: zero offsets (coherent gathers); cfg5l2: 8x32x32 images, B=64 (input of a few images fits L2)
        B, C, O, K, sp = 8, 128, 128, 27, (16, 64, 64)
        if name == "cfg5l2":
            B, sp = 64, (8, 32, 32)
        h = lambda t: t.cuda().half().contiguous()
        x, off, m = h(rn(B, C, *sp)), h(rn(B, 3 * K, *sp)), h(torch.sigmoid(rn(B, K, *sp)))
        if name == "cfg5z":
            off.zero_()
        w = h((torch.rand(O, C, 3, 3, 3, generator=g) * 2 - 1) / math.sqrt(C * K))
        b, go = x.new_empty(0), h(rn(B, O, *sp))
        geo = (3, 3, 3, 1, 1, 1, 2, 2, 2, 2, 2, 2, 1, 1, 64, False)
        out = torch.empty_like(go)
        f = lambda: M.modulated_deform_conv3d_forward_cuda(x, w, b, off, m, out, *geo)
        gi, gw, gb = torch.zeros_like(x), torch.zeros_like(w), torch.zeros_like(b)
        goff, gm = torch.zeros_like(off), torch.zeros_like(m)
        bw = lambda: M.modulated_deform_conv3d_backward_cuda(x, w, b, off, m, gi, gw, gb, goff, gm, go, *geo)
        ns = B * C * K * math.prod(sp)
    tf = timeit(f); pf = _capi.last_path()
    tb = timeit(bw, 2); pb = _capi.last_path()
    print("%s: fwd %.3f ms (%s)  bwd %.3f ms (%s)  -> %.2f GSamples/s" % (name, tf, pf, tb, pb, ns / (tf + tb) / 1e6))


if __name__ == "__main__":
    for n in (sys.argv[1:] or ["cfg4"]):
        run(n)
