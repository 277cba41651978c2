import sys, torch
sys.path.insert(0, ".")
from tests.cases import _c, M2, make_inputs
from tests.util import run_product
from modulated_deform_conv_amd import _capi
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
# cfg3 in fp32: B=32, C=128, O=128, 64x64, G=1? (cfg3: groups 8?) use DG=4
for groups, dg in ((1, 4), (8, 4), (1, 8)):
    case = _c("x", M2, 16, 128, 128, (64, 64), 3, groups=groups, dgroups=dg, bias=False, tier="medium", seed=1)
    tt = make_inputs(case, dtype=torch.float32, device="cuda")
    for path in ("auto", "direct"):
        ms = t(lambda: run_product(case, tt, path))
        print("B16 C128 O128 64x64 G%d DG%d fp32 %s: fwd+bwd %.2f ms (%s)" % (groups, dg, path, ms, run_product(case, tt, path)[2]))
