import math, os, sys
sys.path.insert(0, os.getcwd())
import torch
from modulated_deform_conv_amd import MDCONV_CUDA as M
from tools.bench_configs import timeit
SHAPES = [(4, 256, 256, 56), (4, 64, 64, 56), (4, 128, 128, 56), (3, 256, 256, 56), (2, 256, 256, 56), (16, 256, 256, 28), (12, 128, 256, 28), (8, 64, 256, 40), (1, 256, 256, 100)]
for B, C, O, H in SHAPES:
    g = torch.Generator().manual_seed(0)
    rn = lambda *s: torch.randn(*s, generator=g)
    x, off, m = rn(B, C, H, H), rn(B, 18, H, H), torch.sigmoid(rn(B, 9, H, H))
    w = (torch.rand(O, C, 3, 3, generator=g) * 2 - 1) / math.sqrt(C * 9)
    go = rn(B, O, H, H)
    x, off, m, w, go = [t.cuda().contiguous() for t in (x, off, m, w, go)]
    b = x.new_empty(0)
    geo = (3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 64, False)
    tb = timeit(lambda: M.modulated_deform_conv2d_backward_cuda(x, w, b, off, m, go, *geo), 20)
    print("B=%2d C=%3d O=%3d %dx%d N=%6d: bwd %.3f ms" % (B, C, O, H, H, B*H*H, tb))
