"""Full-shape oracle checks at BASELINE.json configs[1..4] (per-GPU shards):

  cfg2  ModulatedDeformConv2d C=256 56x56 B=32 fp32 (the headline configuration of bench.py)
  cfg3  ModulatedDeformConv2d C=256 56x56 B=32 group=32 deformable_group=4 fp16
  cfg4  DeformConv3d 3x3x3 C=64 32^3 B=8 fp32
  cfg5  ModulatedDeformConv3d C=128 16x64x64 B=8 dilation=2 fp16 (the >2 GiB grad_col path)

The product runs the FULL shard (forward and backward); the oracle is too slow for a whole shard,
but every per-image result (output, grad_input, grad_offset, grad_mask) depends on that image
only (mdeformable_conv.cu:54, 64-66, 228), so image 0 and the last image are compared against the
oracle run on those one-image slices, and EVERY image of the shard against a one-image run of the product.  grad_weight / grad_bias sum over the batch
(mdeformable_conv.cu:436-444): they are checked on the first two images (product B=2 vs oracle
B=2) and, at full size, by additivity over the two half shards.
Tolerances: 1e-4 fp32, 5e-3 fp16 (fp32 oracle on the fp16-rounded inputs), both criteria of
tests.util.assert_close.
"""
import math

import pytest
import torch

import oracle
from tests.util import assert_close

pytestmark = pytest.mark.gpu


def _make(B, C, O, sp, nd, modulated, groups, dgroups, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    K = 3 ** nd
    rn = lambda *s: torch.randn(*s, generator=g)
    x = rn(B, C, *sp)
    off = rn(B, dgroups * nd * K, *sp)
    m = torch.sigmoid(rn(B, dgroups * K, *sp)) if modulated else None
    w = (torch.rand(O, C // groups, *([3] * nd), generator=g) * 2 - 1) / math.sqrt(C * K)
    go = rn(B, O, *sp)
    mv = lambda t: None if t is None else t.to("cuda", dtype).contiguous()
    return mv(x), mv(off), mv(m), mv(w), mv(go)


def _product(op, x, off, m, w, go, geo):
    from modulated_deform_conv_amd import MDCONV_CUDA as M, _capi
    b = x.new_empty(0)
    if op == oracle.MDCN2D:
        out = M.modulated_deform_conv2d_forward_cuda(x, w, b, off, m, *geo)
        gi, goff, gm, gw, _ = M.modulated_deform_conv2d_backward_cuda(x, w, b, off, m, go, *geo)
    else:
        out = torch.empty_like(go)
        gi, gw, goff = torch.empty_like(x), torch.empty_like(w), torch.empty_like(off)
        gb = torch.empty_like(b)
        gm = torch.empty_like(m) if m is not None else None
        with _capi.overwrite_grads():
            if op == oracle.DCN3D:
                M.deform_conv3d_forward_cuda(x, w, b, off, out, *geo)
                M.deform_conv3d_backward_cuda(x, w, b, off, gi, gw, gb, goff, go, *geo)
            else:
                M.modulated_deform_conv3d_forward_cuda(x, w, b, off, m, out, *geo)
                M.modulated_deform_conv3d_backward_cuda(x, w, b, off, m, gi, gw, gb, goff, gm, go, *geo)
    torch.cuda.synchronize()
    return dict(output=out, grad_input=gi, grad_offset=goff, grad_mask=gm, grad_weight=gw), _capi.last_kernels()


def _oracle(op, x, off, m, w, go, pad, dil, groups, dgroups):
    f = lambda t: None if t is None else t.float().cpu()
    out = oracle.forward(op, f(x), f(w), None, f(off), f(m), 1, pad, dil, groups, dgroups, 64, dtype=torch.float32)
    g = oracle.backward(op, f(x), f(w), None, f(off), f(m), f(go), 1, pad, dil, groups, dgroups, 64, dtype=torch.float32)
    g["output"] = out
    return g


def _check_config(op, B, C, O, sp, nd, modulated, groups, dgroups, dtype, pad, dil, tol, seed, want_kernels):
    x, off, m, w, go = _make(B, C, O, sp, nd, modulated, groups, dgroups, dtype, seed)
    k = (3,) * nd
    geo = k + (1,) * nd + (pad,) * nd + (dil,) * nd + (groups, dgroups, 64, False)
    full, kernels = _product(op, x, off, m, w, go, geo)
    assert kernels == want_kernels, kernels
    per_image = ["output", "grad_input", "grad_offset"] + (["grad_mask"] if modulated else [])
    sl = lambda t, s: None if t is None else t[s].contiguous()
    # image 0 and the last image of the full-shard run vs the oracle on the one-image slices
    for img in (0, B - 1):
        s = slice(img, img + 1)
        want = _oracle(op, sl(x, s), sl(off, s), sl(m, s), w, sl(go, s), pad, dil, groups, dgroups)
        for name in per_image:
            assert_close("%s[%d]" % (name, img), full[name][s].float(), want[name], tol)
    # EVERY image of the full-shard run vs a one-image product run (images 0 and B-1 of the full run were just checked
    # against the oracle): a tile-tail, tile-order or batch-chunk-boundary bug in the middle of the batch cannot pass
    for img in range(B):
        s = slice(img, img + 1)
        one, _ = _product(op, sl(x, s), sl(off, s), sl(m, s), w, sl(go, s), geo)
        for name in per_image:
            assert_close("%s[%d] one-image run vs shard" % (name, img), one[name].float(), full[name][s].float(), tol * 0.5)
    # grad_weight on the first two images: product B=2 vs oracle B=2 (the sum over the batch)
    s = slice(0, 2)
    two, _ = _product(op, sl(x, s), sl(off, s), sl(m, s), w, sl(go, s), geo)
    want = _oracle(op, sl(x, s), sl(off, s), sl(m, s), w, sl(go, s), pad, dil, groups, dgroups)
    assert_close("grad_weight[B=2]", two["grad_weight"].float(), want["grad_weight"], tol)
    for name in per_image:   # and the B=2 run reproduces the full run's per-image results
        assert_close("%s[0:2]" % name, two[name].float(), full[name][s].float(), tol * 0.5)
    # full-size grad_weight: additivity over the two half shards (what the multi-GPU all-reduce sums)
    h = B // 2
    parts = [_product(op, sl(x, q), sl(off, q), sl(m, q), w, sl(go, q), geo)[0]["grad_weight"].float()
             for q in (slice(0, h), slice(h, B))]
    assert_close("grad_weight additivity", parts[0] + parts[1], full["grad_weight"].float(), tol)


def test_cfg2_full_shard_vs_oracle():
    _check_config(oracle.MDCN2D, 32, 256, 256, (56, 56), 2, True, 1, 1, torch.float32, 1, 1, 1e-4, 2, "f32")


def test_cfg3_full_shard_vs_oracle():
    _check_config(oracle.MDCN2D, 32, 256, 256, (56, 56), 2, True, 32, 4, torch.float16, 1, 1, 5e-3, 3, "hp")


def test_cfg4_full_shard_vs_oracle():
    _check_config(oracle.DCN3D, 8, 64, 64, (32, 32, 32), 3, False, 1, 1, torch.float32, 1, 1, 1e-4, 4, "f32")


def test_cfg5_full_shard_vs_oracle():
    _check_config(oracle.MDCN3D, 8, 128, 128, (16, 64, 64), 3, True, 1, 1, torch.float16, 2, 2, 5e-3, 5, "hp")
