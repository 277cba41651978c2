import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.cases import CASE_BY_NAME, make_inputs
from tests.util import run_product, run_oracle, rel_err
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2s_mdcn2d_c64_28x28_b4"
case = CASE_BY_NAME[name]
t = make_inputs(case, dtype=torch.float32, device="cuda")
out, g, paths = run_product(case, t, "auto")
_, gd, _ = run_product(case, t, "direct")
wo, w = run_oracle(case, t, torch.float32)
print(paths)
for k in g:
    if g[k] is None: continue
    print(k, "mfma-vs-oracle %.3e" % rel_err(g[k], w[k]), "direct-vs-oracle %.3e" % rel_err(gd[k], w[k]))
go = g["grad_offset"].cpu(); wo_ = w["grad_offset"]
B = go.shape[0]; K = 9
d = (go - wo_).abs().reshape(B, K, 2, -1)
print("per-tap max err", d.amax(dim=(0, 2, 3)))
print("per-batch max err", d.amax(dim=(1, 2, 3)))
e = d.amax(dim=(0,1,2))
print("per-pixel err: first 64", (e[:64] > 1e-3).int().tolist())
print("ratio sample", (go.flatten()[:8] / wo_.flatten()[:8]).tolist())
print("---- input = ones ----")
t["input"] = torch.ones_like(t["input"])
out, g, paths = run_product(case, t, "auto")
wo, w = run_oracle(case, t, torch.float32)
for k in ("grad_offset", "grad_mask"):
    print(k, "mfma-vs-oracle %.3e" % rel_err(g[k], w[k]))
print("---- input = channel index ----")
C = t["input"].shape[1]
t["input"] = torch.arange(C, device="cuda", dtype=torch.float32).view(1, C, 1, 1).expand_as(t["input"]).contiguous()
out, g, paths = run_product(case, t, "auto")
wo, w = run_oracle(case, t, torch.float32)
for k in ("grad_offset", "grad_mask"):
    print(k, "mfma-vs-oracle %.3e" % rel_err(g[k], w[k]))
print((g["grad_mask"].flatten()[:6]).tolist(), w["grad_mask"].flatten()[:6].tolist())
