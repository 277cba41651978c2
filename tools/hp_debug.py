#!/usr/bin/env python3
"""Developer aid: per-tensor errors of the 16-bit path vs the fp32 oracle for named cases."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.cases import CASE_BY_NAME, make_inputs
from tests.test_gpu_hp import HP_CASES
from tests.util import run_oracle, run_product, rel_err, elem_err
from modulated_deform_conv_amd import _capi
allc = dict(CASE_BY_NAME); allc.update({c["name"]: c for c in HP_CASES})
names = [a for a in sys.argv[1:] if not a.startswith("-")] or [c["name"] for c in HP_CASES]
dt = torch.bfloat16 if "--bf16" in sys.argv else torch.float16
for n in names:
    case = allc[n]
    t = make_inputs(case, dtype=dt, device="cuda")
    try:
        out, grads, paths = run_product(case, t, "auto")
        torch.cuda.synchronize()
    except Exception as e:
        print(n, "FAILED", e); continue
    wo, w = run_oracle(case, {k: (None if v is None else v.float()) for k, v in t.items()}, torch.float32)
    msg = ["out %.1e/%.1e" % (rel_err(out.float(), wo), elem_err(out.float(), wo))]
    for k, g in grads.items():
        if w[k] is not None:
            msg.append("%s %.1e/%.1e" % (k[5:], rel_err(g.float(), w[k]), elem_err(g.float(), w[k])))
    print("%-34s %s %s" % (n, _capi.last_kernels(), "  ".join(msg)))
