#!/usr/bin/env python3
"""Developer aid: forward time of small fp32 MDCN2d calls (few tiles: the tap-range plan of mfma_fwd.hip below one
dispatch round), graph-replayed so that host launch latency is out of the loop.
usage: [MDCONV_FWD_TAIL=1] python tools/small_fwd.py"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from modulated_deform_conv_amd import MDCONV_CUDA as M  # noqa: E402

for B, C, O, H in ((1, 256, 256, 14), (1, 256, 256, 28), (1, 256, 256, 56), (2, 256, 256, 56), (8, 64, 64, 28), (2, 128, 128, 56)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, C, H, H, generator=g).cuda()
    off = torch.randn(B, 18, H, H, generator=g).cuda()
    m = torch.sigmoid(torch.randn(B, 9, H, H, generator=g)).cuda()
    w = ((torch.rand(O, C, 3, 3, generator=g) * 2 - 1) / math.sqrt(C * 9)).cuda()
    b = torch.zeros(O).cuda()
    geo = (3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 64, True)
    f = lambda: M.modulated_deform_conv2d_forward_cuda(x, w, b, off, m, *geo)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        f(); f()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        f()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    print("B=%d C=%d O=%d %dx%d: forward %.1f us (graph replay)" % (B, C, O, H, H, e0.elapsed_time(e1) / 50 * 1e3), flush=True)
