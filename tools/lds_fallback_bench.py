import os, sys, torch
sys.path.insert(0, '/root/repo')
from tests.cases import _c, make_inputs, M3, D3
from tests.util import run_product
from modulated_deform_conv_amd import _capi
# a realistic channel-expanding 3-D layer: C_in = 64 -> C_out = 256, 16^3, B = 4
case = _c("bench", M3, 4, 64, 256, (16, 16, 16), 3, tier="medium", seed=1)
t = make_inputs(case, device="cuda")
for path in ("auto", "direct"):
    for _ in range(2): out, g, p = run_product(case, t, path)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): out, g, p = run_product(case, t, path)
    e1.record(); torch.cuda.synchronize()
    print(path, p, "%.3f ms per fwd+bwd" % (e0.elapsed_time(e1) / 5))
