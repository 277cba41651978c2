python - <<PY
import torch, sys
sys.path.insert(0, ".")
from modulated_deform_conv_amd import modulated_deform_conv as mdc
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for (C, B, S, dt) in ((256, 32, 56, torch.float32), (256, 32, 56, torch.float16), (64, 4, 28, torch.float32), (64, 4, 28, torch.float16), (128, 8, 40, torch.float32)):
    m = mdc.ModulatedDeformConv2dPack(C, C, 3, padding=1, bias=True).cuda().to(dt)
    x = torch.randn(B, C, S, S, device="cuda", dtype=dt)
    co, cm = m.conv_offset, m.conv_mask
    def fused():
        y = torch.nn.functional.conv2d(x, torch.cat((co.weight, cm.weight)), torch.cat((co.bias, cm.bias)), 1, 1)
        return y[:, :18].contiguous(), y[:, 18:].contiguous()
    with torch.no_grad():
        print(C, B, S, dt, "own %.3f  two-convs %.3f  fused-framework %.3f ms" % (t(lambda: m._side(x)), t(lambda: (co(x), cm(x))), t(fused)))
PY
