#!/usr/bin/env python3
"""Developer aid: forward / backward time of fp16 MDCN2d 56x56 C=O=256 B=32 for (groups, dgroups) variants."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modulated_deform_conv_amd import MDCONV_CUDA as M, _capi
from tools.bench_configs import timeit
g = torch.Generator().manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g)
B, C, O, K, sp = 32, 256, 256, 9, (56, 56)
for G, DG in ((32, 4), (32, 1), (1, 4), (1, 1), (8, 2)):
    h = lambda t: t.cuda().half().contiguous()
    x, off, m = h(rn(B, C, *sp)), h(rn(B, DG * 2 * K, *sp)), h(torch.sigmoid(rn(B, DG * K, *sp)))
    w = h((torch.rand(O, C // G, 3, 3, generator=g) * 2 - 1) / math.sqrt(C * K))
    b, go = x.new_empty(0), h(rn(B, O, *sp))
    geo = (3, 3, 1, 1, 1, 1, 1, 1, G, DG, 64, False)
    tf = timeit(lambda: M.modulated_deform_conv2d_forward_cuda(x, w, b, off, m, *geo))
    tb = timeit(lambda: M.modulated_deform_conv2d_backward_cuda(x, w, b, off, m, go, *geo), 3)
    print("G=%2d DG=%d: fwd %.3f ms  bwd %.3f ms (%s)" % (G, DG, tf, tb, _capi.last_kernels()))
