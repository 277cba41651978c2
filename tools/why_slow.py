#!/usr/bin/env python3
"""Developer aid: per-kernel times (library events) + kernel trace names of one shape.  usage: why_slow.py op dtype B C O size.. -- DG"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.cases import D2, D3, M2, M3, _c, make_inputs
from tests.util import run_product
from modulated_deform_conv_amd import _capi
op = {"D2": D2, "D3": D3, "M2": M2, "M3": M3}[sys.argv[1]]
dtype = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[sys.argv[2]]
B, C, O = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
i = sys.argv.index("--")
size = tuple(int(v) for v in sys.argv[6:i])
dg = int(sys.argv[i + 1])
case = _c("w", op, B, C, O, size, 3, dgroups=dg, tier="medium", seed=1)
t = make_inputs(case, dtype=dtype, device="cuda")
for _ in range(2):
    run_product(case, t, "auto")
torch.cuda.synchronize()
_capi.profile_enable(True); _capi.profile_reset()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    _, _, p = run_product(case, t, "auto")
e1.record(); torch.cuda.synchronize()
_capi.profile_enable(False)
print(sys.argv[1:], p, "%.3f ms" % (e0.elapsed_time(e1) / 3), {k: round(v[1], 3) for k, v in _capi.profile_read().items()})
