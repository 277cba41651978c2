#!/usr/bin/env python3
"""Developer aid: graph-replayed forward time of small channels-last (3-D / narrow 2-D) fp32 forwards -- the shapes whose
tile grid is smaller than one dispatch round.   MDCONV_FWD_TAIL=0 = no tap-range split (the round-4 behaviour of this kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.cases import D3, M3, M2, _c, make_inputs
from tests.util import tup
from modulated_deform_conv_amd import MDCONV_CUDA as M
SH = [("M3", M3, 4, 256, 256, (4, 7, 7)), ("M3", M3, 4, 128, 128, (4, 14, 14)), ("M3", M3, 2, 64, 64, (8, 28, 28)), ("D3", D3, 4, 64, 128, (8, 14, 14)),
      ("M3", M3, 1, 64, 64, (16, 16, 16)), ("M3", M3, 8, 64, 64, (32, 32, 32)), ("M2", M2, 8, 64, 64, (56, 56)), ("M2", M2, 2, 128, 128, (56, 56))]
for name, op, B, C, O, sz in SH:
    case = _c(name, op, B, C, O, sz, 3, tier="medium", seed=1)
    t = make_inputs(case, device="cuda")
    nd = len(sz)
    k, s, p, d = (tup(case[x], nd) for x in ("k", "stride", "padding", "dilation"))
    geo = k + s + p + d + (1, 1, 64, True)
    x, w, off, m, b = t["input"], t["weight"], t["offset"], t["mask"], t["bias"]
    out = torch.empty_like(t["grad_output"])
    def fwd():
        if op == M3: M.modulated_deform_conv3d_forward_cuda(x, w, b, off, m, out, *geo)
        elif op == D3: M.deform_conv3d_forward_cuda(x, w, b, off, out, *geo)
        else: return M.modulated_deform_conv2d_forward_cuda(x, w, b, off, m, *geo)
    for _ in range(3): fwd()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr): fwd()
    for _ in range(3): gr.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): gr.replay()
    e1.record(); torch.cuda.synchronize()
    print("%s B=%d C=%d O=%d %-12s %8.1f us" % (name, B, C, O, "x".join(map(str, sz)), 1e3 * e0.elapsed_time(e1) / 20), flush=True)
