#!/usr/bin/env python3
"""Developer aid: fp32 ModulatedDeformConv2d forward / backward times over a few shapes (run it under
different MDCONV_FWD_CL / MDCONV_BWD_CL / MDCONV_BD_CL settings to compare kernel choices)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modulated_deform_conv_amd import MDCONV_CUDA as M
from tools.bench_configs import timeit

SHAPES = [(32, 64, 64, 56), (32, 128, 128, 56), (32, 256, 256, 28), (8, 256, 256, 56), (16, 512, 512, 28),
          (32, 256, 256, 56), (32, 128, 256, 56), (32, 256, 128, 56), (4, 256, 256, 56)]
for B, C, O, H in SHAPES:
    g = torch.Generator().manual_seed(0)
    rn = lambda *s: torch.randn(*s, generator=g)
    x, off, m = rn(B, C, H, H), rn(B, 18, H, H), torch.sigmoid(rn(B, 9, H, H))
    w = (torch.rand(O, C, 3, 3, generator=g) * 2 - 1) / math.sqrt(C * 9)
    go = rn(B, O, H, H)
    x, off, m, w, go = [t.cuda().contiguous() for t in (x, off, m, w, go)]
    b = x.new_empty(0)
    geo = (3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 64, False)
    tf = timeit(lambda: M.modulated_deform_conv2d_forward_cuda(x, w, b, off, m, *geo), 5)
    tb = timeit(lambda: M.modulated_deform_conv2d_backward_cuda(x, w, b, off, m, go, *geo), 5)
    print("B=%2d C=%3d O=%3d %dx%d: fwd %.3f ms  bwd %.3f ms" % (B, C, O, H, H, tf, tb))
