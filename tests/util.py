"""Helpers shared by the GPU parity tests."""
import torch

import oracle
from tests.cases import D2, D3, M2, M3, ndim


def tup(v, nd):
    return (v,) * nd if isinstance(v, int) else tuple(v)


def run_product(case, t, path=None):
    """Run forward + backward of ``case`` through the MDCONV_CUDA surface (C ABI underneath).

    ``t`` holds device tensors from make_inputs().  Returns (output, grads-dict, paths)."""
    from modulated_deform_conv_amd import MDCONV_CUDA as M
    from modulated_deform_conv_amd import _capi
    op, nd = case["op"], ndim(case)
    k, s, p, d = (tup(case[x], nd) for x in ("k", "stride", "padding", "dilation"))
    geo = k + s + p + d + (case["groups"], case["dgroups"], case["in_step"], case["bias"])
    x, w, off, m, go = t["input"], t["weight"], t["offset"], t["mask"], t["grad_output"]
    b = t["bias"] if case["bias"] else x.new_empty(0)
    prev = _capi.set_path(path) if path else None
    paths = []
    try:
        if op == M2:
            out = M.modulated_deform_conv2d_forward_cuda(x, w, b, off, m, *geo)
            paths.append(_capi.last_path())
            gi, goff, gm, gw, gb = M.modulated_deform_conv2d_backward_cuda(x, w, b, off, m, go, *geo)
            paths.append(_capi.last_path())
        else:
            out = torch.empty_like(go)
            gi = torch.zeros_like(x, memory_format=torch.contiguous_format)
            gw, goff = torch.zeros_like(w), torch.zeros_like(off)
            gb = torch.zeros_like(b)
            gm = torch.zeros_like(m) if m is not None else None
            if op == D2:
                M.deform_conv2d_forward_cuda(x, w, b, off, out, *geo)
                paths.append(_capi.last_path())
                M.deform_conv2d_backward_cuda(x, w, b, off, gi, gw, gb, goff, go, *geo)
            elif op == D3:
                M.deform_conv3d_forward_cuda(x, w, b, off, out, *geo)
                paths.append(_capi.last_path())
                M.deform_conv3d_backward_cuda(x, w, b, off, gi, gw, gb, goff, go, *geo)
            else:
                M.modulated_deform_conv3d_forward_cuda(x, w, b, off, m, out, *geo)
                paths.append(_capi.last_path())
                M.modulated_deform_conv3d_backward_cuda(x, w, b, off, m, gi, gw, gb, goff, gm, go, *geo)
            paths.append(_capi.last_path())
    finally:
        if prev is not None:
            _capi.set_path(prev)
    grads = dict(grad_input=gi, grad_offset=goff, grad_mask=gm, grad_weight=gw,
                 grad_bias=gb if case["bias"] else None)
    return out, grads, paths


def run_oracle(case, t, dtype, intermediates=None):
    """Oracle forward + backward on CPU copies of ``t`` computed in ``dtype``; ``intermediates`` = a 16-bit torch dtype
    rounds the reference's columns / grad_columns buffers to it where the reference stores them (oracle.backward)."""
    args = (case["stride"], case["padding"], case["dilation"], case["groups"], case["dgroups"],
            case["in_step"])
    c = {k: (None if v is None else v.detach().cpu()) for k, v in t.items()}
    out = oracle.forward(case["op"], c["input"], c["weight"], c["bias"], c["offset"], c["mask"],
                         *args, dtype=dtype)
    g = oracle.backward(case["op"], c["input"], c["weight"], c["bias"], c["offset"], c["mask"],
                        c["grad_output"], *args, dtype=dtype, intermediates=intermediates)
    return out, g


def rel_err(a, b):
    """max|a-b| / max(1, max|b|) -- the 'relative to scale' reading of the 1e-4 fp32 tolerance
    (SURVEY.md section 8c tolerance note)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / max(1.0, b.abs().max().item())).item()


def elem_err(a, b):
    """Per-element criterion: max over elements of |a-b| / (rms(b) + |b|).  Unlike rel_err it does
    not let small elements of a large-magnitude tensor (grad_weight at cfg2: |max| ~ 3e2) hide
    behind the tensor's maximum: the absolute slack is tol * rms(b), not tol * max|b|."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    if b.numel() == 0:
        return 0.0
    rms = b.pow(2).mean().sqrt().item()
    return ((a - b).abs() / (max(rms, 1e-30) + b.abs())).max().item()


WORST = {}   # name -> worst (scaled max error, per-element error) seen in this process


def assert_close(name, got, want, tol, elem_tol=None):
    """Both criteria must hold: scaled max error <= tol and per-element error <= elem_tol
    (default: the same tol)."""
    assert got.shape == want.shape, (name, got.shape, want.shape)
    assert torch.isfinite(got.float()).all(), "%s: non-finite values" % name
    e = rel_err(got, want)
    pe = elem_err(got, want)
    w = WORST.get(name, (0.0, 0.0))
    WORST[name] = (max(w[0], e), max(w[1], pe))
    assert e <= tol, "%s: scaled max error %.3e > %.1e" % (name, e, tol)
    et = tol if elem_tol is None else elem_tol
    assert pe <= et, "%s: per-element error %.3e > %.1e (scaled max error %.3e)" % (name, pe, et, e)
