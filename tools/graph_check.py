#!/usr/bin/env python3
"""Capture forward + backward of the cfg2 layer into a HIP graph (torch.cuda.graph) and replay it:
checks that every launch of the C ABI is capturable and times eager vs replay."""
import math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modulated_deform_conv_amd import MDCONV_CUDA as M

B, C, O, H, W, K = int(os.environ.get("GC_B", 32)), 256, 256, int(os.environ.get("GC_H", 56)), int(os.environ.get("GC_H", 56)), 9
g = torch.Generator().manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g).cuda()
x, off, m = rn(B, C, H, W), rn(B, 18, H, W), torch.sigmoid(rn(B, 9, H, W))
w = ((torch.rand(O, C, 3, 3, generator=g) * 2 - 1) / math.sqrt(C * K)).cuda()
b, go = (0.1 * torch.randn(O, generator=g)).cuda(), rn(B, O, H, W)
geo = (3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 64, True)

WHICH = os.environ.get("GC_WHICH", "both")
def step():
    r = ()
    if WHICH in ("both", "fwd"):
        r += (M.modulated_deform_conv2d_forward_cuda(x, w, b, off, m, *geo),)
    if WHICH in ("both", "bwd"):
        r += tuple(M.modulated_deform_conv2d_backward_cuda(x, w, b, off, m, go, *geo))
    return r

def timeit(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

ref = step(); torch.cuda.synchronize()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(s)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    res = step()
graph.replay(); torch.cuda.synchronize()
ok = all(torch.allclose(a, r, rtol=1e-5, atol=1e-5 * max(1.0, r.abs().max().item())) for a, r in zip(res, ref))
print("graph replay matches eager:", ok)
print("eager  %.3f ms/step" % timeit(step))
print("replay %.3f ms/step" % timeit(graph.replay))
