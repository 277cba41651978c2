"""CPU tests that pin the oracle (no GPU):

* the my_test.py scenario's known answers (reference my_test.py:1-35; values SURVEY.md section 4),
* zero offsets + unit mask == F.conv2d / F.conv3d for output and gradients,
* agreement with an independent PyTorch-autograd statement of the operator (tests/torch_ref.py),
* fp64 central finite differences,
* in_step invariance (the reference's modulated-2D op is in_step-invariant).
"""
import itertools

import pytest
import torch
import torch.nn.functional as F

import oracle
from tests.cases import CASES, _c, make_inputs
from tests.torch_ref import deform_conv_nd

ALL_OPS = (oracle.DCN2D, oracle.MDCN2D, oracle.DCN3D, oracle.MDCN3D)


def _nd(op):
    return 3 if op in (oracle.DCN3D, oracle.MDCN3D) else 2


def _mod(op):
    return op in (oracle.MDCN2D, oracle.MDCN3D)


# ------------------------------------------------------------------ my_test.py known answers
def test_my_test_scenario_known_answers(oracle_lib):
    x = torch.ones(1, 1, 5, 5)
    offset = torch.zeros(1, 18, 5, 5)
    mask = torch.ones(1, 9, 5, 5)
    weight = torch.ones(1, 1, 3, 3)
    bias = torch.zeros(1)
    counts = torch.tensor([[4, 6, 6, 6, 4], [6, 9, 9, 9, 6], [6, 9, 9, 9, 6], [6, 9, 9, 9, 6],
                           [4, 6, 6, 6, 4]], dtype=torch.float32)
    for op in (oracle.DCN2D, oracle.MDCN2D):
        out = oracle.forward(op, x, weight, bias, offset, mask, 1, 1, 1, 1, 1, 64)
        assert torch.equal(out[0, 0], counts)
        assert out.sum().item() == 169
        g = oracle.backward(op, x, weight, bias, offset, mask, torch.ones_like(out), 1, 1, 1, 1, 1, 64)
        assert torch.equal(g["grad_input"][0, 0], counts)
        assert torch.equal(g["grad_weight"].flatten(),
                           torch.tensor([16., 20, 16, 20, 25, 20, 16, 20, 16]))
        assert torch.equal(g["grad_bias"], torch.tensor([25.]))
        if op == oracle.MDCN2D:
            gm0 = torch.zeros(5, 5)
            gm0[1:, 1:] = 1
            assert torch.equal(g["grad_mask"][0, 0], gm0)
            # right-sided differences at the border (quirk Q2): sum |grad_offset| = 52
            assert g["grad_offset"].abs().sum().item() == 52


# ------------------------------------------------------------------ zero offset == plain conv
@pytest.mark.parametrize("op", ALL_OPS)
@pytest.mark.parametrize("stride,dil,groups,dg,in_step", [(1, 1, 1, 1, 64), (2, 1, 2, 2, 1),
                                                           (1, 2, 1, 4, 2), (2, 2, 4, 1, 3)])
def test_zero_offset_equals_conv(oracle_lib, op, stride, dil, groups, dg, in_step):
    torch.manual_seed(1)
    nd = _nd(op)
    B, C, O = 2, 8, 4
    in_sz = (7, 6) if nd == 2 else (5, 6, 4)
    k = 3
    pad = dil
    x = torch.randn(B, C, *in_sz, dtype=torch.float64)
    w = torch.randn(O, C // groups, *([k] * nd), dtype=torch.float64)
    b = torch.randn(O, dtype=torch.float64)
    conv = F.conv2d if nd == 2 else F.conv3d
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    ref = conv(xr, wr, br, stride, pad, dil, groups)
    K = k ** nd
    offset = torch.zeros(B, dg * nd * K, *ref.shape[2:], dtype=torch.float64)
    mask = torch.ones(B, dg * K, *ref.shape[2:], dtype=torch.float64)
    out = oracle.forward(op, x, w, b, offset, mask, stride, pad, dil, groups, dg, in_step)
    assert torch.allclose(out, ref, rtol=1e-12, atol=1e-12)
    go = torch.randn_like(ref)
    ref.backward(go)
    g = oracle.backward(op, x, w, b, offset, mask, go, stride, pad, dil, groups, dg, in_step)
    assert torch.allclose(g["grad_input"], xr.grad, rtol=1e-11, atol=1e-11)
    assert torch.allclose(g["grad_weight"], wr.grad, rtol=1e-11, atol=1e-11)
    assert torch.allclose(g["grad_bias"], br.grad, rtol=1e-11, atol=1e-11)


# ------------------------------------------------------------------ vs independent autograd ref
@pytest.mark.parametrize("case", [c for c in CASES if c["tier"] == "small"], ids=lambda c: c["name"])
def test_oracle_matches_autograd_reference(oracle_lib, case):
    t = make_inputs(case, dtype=torch.float64)
    op = case["op"]
    args = (case["stride"], case["padding"], case["dilation"], case["groups"], case["dgroups"])
    leaves = {k: t[k].clone().requires_grad_() for k in ("input", "offset", "weight")}
    if t["mask"] is not None:
        leaves["mask"] = t["mask"].clone().requires_grad_()
    if t["bias"] is not None:
        leaves["bias"] = t["bias"].clone().requires_grad_()
    ref = deform_conv_nd(leaves["input"], leaves["offset"], leaves.get("mask"), leaves["weight"],
                         leaves.get("bias"), *args)
    out = oracle.forward(op, t["input"], t["weight"], t["bias"], t["offset"], t["mask"], *args,
                         case["in_step"])
    assert torch.allclose(out, ref, rtol=1e-11, atol=1e-11)
    go = torch.randn_like(ref)
    ref.backward(go)
    g = oracle.backward(op, t["input"], t["weight"], t["bias"], t["offset"], t["mask"], go, *args,
                        case["in_step"])
    for name, key in (("input", "grad_input"), ("offset", "grad_offset"), ("mask", "grad_mask"),
                      ("weight", "grad_weight"), ("bias", "grad_bias")):
        if name in leaves:
            assert torch.allclose(g[key], leaves[name].grad, rtol=1e-10, atol=1e-10), key


# ------------------------------------------------------------------ fp64 finite differences
@pytest.mark.parametrize("op", ALL_OPS)
def test_finite_differences(oracle_lib, op):
    torch.manual_seed(3)
    nd = _nd(op)
    B, C, O, groups, dg = 2, 4, 4, 2, 2
    in_sz = (5, 4) if nd == 2 else (4, 3, 4)
    k, stride, pad, dil, in_step = 2, 1, 1, 1, 1
    K = k ** nd
    x = torch.randn(B, C, *in_sz, dtype=torch.float64)
    w = torch.randn(O, C // groups, *([k] * nd), dtype=torch.float64)
    b = torch.randn(O, dtype=torch.float64)
    out_sz = tuple((n + 2 * pad - (dil * (k - 1) + 1)) // stride + 1 for n in in_sz)
    offset = torch.randn(B, dg * nd * K, *out_sz, dtype=torch.float64) * 0.7 + 0.013
    mask = torch.sigmoid(torch.randn(B, dg * K, *out_sz, dtype=torch.float64)) if _mod(op) else None
    args = (stride, pad, dil, groups, dg, in_step)
    go = torch.randn(B, O, *out_sz, dtype=torch.float64)

    def loss(x_, w_, b_, off_, m_):
        return (oracle.forward(op, x_, w_, b_, off_, m_, *args) * go).sum().item()

    g = oracle.backward(op, x, w, b, offset, mask, go, *args)
    eps = 1e-6
    gen = torch.Generator().manual_seed(0)
    tensors = dict(grad_input=x, grad_weight=w, grad_bias=b, grad_offset=offset)
    if mask is not None:
        tensors["grad_mask"] = mask
    for key, tens in tensors.items():
        flat = tens.view(-1)
        for i in torch.randperm(flat.numel(), generator=gen)[:12].tolist():
            keep = flat[i].item()
            flat[i] = keep + eps
            lp = loss(x, w, b, offset, mask)
            flat[i] = keep - eps
            lm = loss(x, w, b, offset, mask)
            flat[i] = keep
            fd = (lp - lm) / (2 * eps)
            assert abs(fd - g[key].view(-1)[i].item()) < 1e-6 * max(1.0, abs(fd)), (key, i)


# ------------------------------------------------------------------ in_step invariance
@pytest.mark.parametrize("op", ALL_OPS)
def test_in_step_invariance(oracle_lib, op):
    case = _c("in_step", op, 4, 4, 4, (6, 5) if _nd(op) == 2 else (4, 5, 3), 3, dgroups=2, seed=11)
    t = make_inputs(case, dtype=torch.float64)
    args = (1, 1, 1, 1, 2)
    outs, grads = [], []
    for in_step in (1, 2, 3, 4, 64):
        outs.append(oracle.forward(op, t["input"], t["weight"], t["bias"], t["offset"], t["mask"],
                                   *args, in_step))
        grads.append(oracle.backward(op, t["input"], t["weight"], t["bias"], t["offset"], t["mask"],
                                     t["grad_output"], *args, in_step))
    for o, g in zip(outs[1:], grads[1:]):
        assert torch.allclose(o, outs[0], rtol=1e-13, atol=1e-13)
        for key in g:
            if g[key] is not None:
                assert torch.allclose(g[key], grads[0][key], rtol=1e-12, atol=1e-12), key


def test_shape_errors(oracle_lib):
    x = torch.randn(2, 4, 5, 5)
    w = torch.randn(4, 4, 3, 3)
    off = torch.zeros(2, 18, 5, 5)
    with pytest.raises(RuntimeError):
        oracle.forward(oracle.DCN2D, x, w, None, off, None, 1, 1, 1, 3, 1, 64)   # C % groups
    with pytest.raises(RuntimeError):
        oracle.forward(oracle.DCN2D, x, w, None, off, None, 1, 1, 1, 1, 1, 0)    # in_step = 0


@pytest.mark.parametrize("inter", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_intermediate_storage_rounding_sits_where_the_reference_stores_them(oracle_lib, inter):
    """`oracle.backward(..., intermediates=half)` rounds grad_columns after GEMM-1 and columns before GEMM-2 -- the two
    buffers the reference allocates with input.options() (mdeformable_conv.cu:396-397) -- and nothing else.  1 x 1 kernel,
    zero offsets, so that both are simple products: grad_mask = sum_c round(grad_col[c]) x[c], grad_weight[o, c] =
    sum_n g[o, n] round(mask x[c, n]); the rounding itself is checked against torch's conversion (subnormals included)."""
    g = torch.Generator().manual_seed(5)
    B, C, O, H, W = 2, 5, 3, 4, 3
    x = torch.randn(B, C, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(O, C, 1, 1, generator=g, dtype=torch.float64) * torch.tensor([1.0, 1e-3, 3e-6]).view(O, 1, 1, 1)
    off = torch.zeros(B, 2, H, W, dtype=torch.float64)
    m = torch.rand(B, 1, H, W, generator=g, dtype=torch.float64)
    go = torch.randn(B, O, H, W, generator=g, dtype=torch.float64)
    got = oracle.backward(oracle.MDCN2D, x, w, None, off, m, go, 1, 0, 1, 1, 1, 64, dtype=torch.float64, intermediates=inter)
    rnd = lambda t: t.float().to(inter).double()
    gcol = rnd(torch.einsum("oc,bohw->bchw", w[:, :, 0, 0], go))
    want_gm = (gcol * x).sum(1, keepdim=True)
    col = rnd(x * m)
    want_gw = torch.einsum("bohw,bchw->oc", go, col)[:, :, None, None]
    assert torch.allclose(got["grad_mask"], want_gm, rtol=1e-12, atol=1e-14)
    assert torch.allclose(got["grad_weight"], want_gw, rtol=1e-12, atol=1e-14)
    plain = oracle.backward(oracle.MDCN2D, x, w, None, off, m, go, 1, 0, 1, 1, 1, 64, dtype=torch.float64)
    assert not torch.equal(plain["grad_mask"], got["grad_mask"])          # the mode does something ...
    again = oracle.backward(oracle.MDCN2D, x, w, None, off, m, go, 1, 0, 1, 1, 1, 64, dtype=torch.float64)
    assert torch.equal(plain["grad_mask"], again["grad_mask"])            # ... and is reset after the call
