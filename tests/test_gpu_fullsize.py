"""Full-size GPU tests at BASELINE.json's configurations, through size-independent properties
(the oracle would take minutes at these sizes):

  * two independent implementations agree: MFMA path vs direct (VALU) path, same inputs;
  * zero offsets + unit mask == F.conv2d / F.conv3d (rocm convolution as the independent check);
  * linearity in the weights; batch-shard equivalence (what the multi-GPU path relies on);
  * in_step invariance and run-to-run determinism of the forward.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from tests.util import assert_close

pytestmark = pytest.mark.gpu


def _inputs(B, C, O, spatial, K, nd, modulated, dtype=torch.float32, seed=0):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    x = rn(B, C, *spatial)
    off = rn(B, nd * K, *spatial)
    m = torch.sigmoid(rn(B, K, *spatial)) if modulated else None
    w = (torch.rand(O, C, *([3] * nd), generator=g) * 2 - 1) / math.sqrt(C * K)
    b = 0.1 * rn(O)
    go = rn(B, O, *spatial)
    mv = lambda t: None if t is None else t.to("cuda", dtype).contiguous()
    return mv(x), mv(off), mv(m), mv(w), mv(b), mv(go)


def _mdcn2d(x, off, m, w, b, go, path, in_step=64):
    from modulated_deform_conv_amd import MDCONV_CUDA as M, _capi
    geo = (3, 3, 1, 1, 1, 1, 1, 1, 1, 1, in_step, True)
    prev = _capi.set_path(path)
    try:
        out = M.modulated_deform_conv2d_forward_cuda(x, w, b, off, m, *geo)
        p1 = _capi.last_path()
        grads = M.modulated_deform_conv2d_backward_cuda(x, w, b, off, m, go, *geo) if go is not None else None
        p2 = _capi.last_path()
    finally:
        _capi.set_path(prev)
    return out, grads, (p1, p2)


# --------------------------------------------------------------------------- cfg2 (headline)
@pytest.fixture(scope="module")
def cfg2():
    return _inputs(32, 256, 256, (56, 56), 9, 2, True)


def test_cfg2_mfma_equals_direct(cfg2):
    """ModulatedDeformConv2d 3x3 C=256 56x56 B=32 fp32: the two kernel paths agree on the output and
    on all five gradients within the fp32 parity tolerance."""
    x, off, m, w, b, go = cfg2
    out_a, g_a, paths = _mdcn2d(x, off, m, w, b, go, "mfma")
    assert paths == ("mfma", "mfma")
    out_d, g_d, paths = _mdcn2d(x, off, m, w, b, go, "direct")
    assert paths == ("direct", "direct")
    assert_close("output", out_a, out_d, 1e-4)
    for name, a, d in zip(("grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias"), g_a, g_d):
        assert_close(name, a, d, 1e-4)


def test_cfg2_zero_offset_equals_conv2d(cfg2):
    x, off, m, w, b, go = cfg2
    out, g, _ = _mdcn2d(x, torch.zeros_like(off), torch.ones_like(m), w, b, go, "auto")
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    ref = F.conv2d(xr, wr, br, 1, 1)
    ref.backward(go)
    assert_close("output", out, ref, 1e-4)
    assert_close("grad_input", g[0], xr.grad, 1e-4)
    assert_close("grad_weight", g[3], wr.grad, 1e-4)
    assert_close("grad_bias", g[4], br.grad, 1e-4)


def test_cfg2_linearity_shards_instep_determinism(cfg2):
    x, off, m, w, b, go = cfg2
    out, g, _ = _mdcn2d(x, off, m, w, b, go, "auto")
    # determinism + in_step invariance of the forward (bitwise)
    out2, _, _ = _mdcn2d(x, off, m, w, b, None, "auto", in_step=1)
    assert torch.equal(out, out2)
    # linearity in the weights (bias off: pass zeros)
    zb = torch.zeros_like(b)
    w2 = torch.randn_like(w) * w.std()
    o1, _, _ = _mdcn2d(x, off, m, w, zb, None, "auto")
    o2, _, _ = _mdcn2d(x, off, m, w2, zb, None, "auto")
    o12, _, _ = _mdcn2d(x, off, m, w + w2, zb, None, "auto")
    assert_close("linearity", o12, o1 + o2, 1e-4)
    # batch shards: per-image results identical, weight/bias gradients add up (multi-GPU contract)
    h = x.shape[0] // 2
    parts = [_mdcn2d(x[s].contiguous(), off[s].contiguous(), m[s].contiguous(), w, b, go[s].contiguous(), "auto")
             for s in (slice(0, h), slice(h, None))]
    # (to fp32 re-association: the tiles of the forward's last dispatch round are summed over tap ranges,
    # mfma_fwd.hip, and which tiles those are depends on the batch size -- SURVEY.md section 8e asks for
    # "fp32 reassociation tolerance" between shards and the single-GPU run, not for equal bits)
    assert_close("output", torch.cat([p[0] for p in parts]), out, 1e-5)
    assert_close("grad_input", torch.cat([p[1][0] for p in parts]), g[0], 1e-5)
    assert_close("grad_offset", torch.cat([p[1][1] for p in parts]), g[1], 1e-5)
    assert_close("grad_mask", torch.cat([p[1][2] for p in parts]), g[2], 1e-5)
    assert_close("grad_weight", parts[0][1][3] + parts[1][1][3], g[3], 1e-4)
    assert_close("grad_bias", parts[0][1][4] + parts[1][1][4], g[4], 1e-4)


@pytest.mark.parametrize("nb", [4, 8])
def test_cfg2_small_shards_match_the_full_batch(cfg2, nb):
    """The strong-scaling shards (4 / 8 images of cfg2's 32): their forward runs ONE dispatch round made of tap-range
    workgroups of mixed length (mfma_fwd.hip, fwd_tail_plan: 392 tiles -> 240 x 3 + 152 x 2 ranges; 784 tiles -> 544 whole +
    240 x 2), a plan no other test shape reaches.  Per-image results equal the full-batch run's to fp32 re-association."""
    x, off, m, w, b, go = cfg2
    out, g, _ = _mdcn2d(x, off, m, w, b, go, "auto")
    s = slice(8, 8 + nb)
    o2, g2, _ = _mdcn2d(x[s].contiguous(), off[s].contiguous(), m[s].contiguous(), w, b, go[s].contiguous(), "auto")
    assert_close("output", o2, out[s], 1e-5)
    assert_close("grad_input", g2[0], g[0][s], 1e-5)
    assert_close("grad_offset", g2[1], g[1][s], 1e-5)
    assert_close("grad_mask", g2[2], g[2][s], 1e-5)


BITSTABLE_CODE = r"""
import sys
sys.path.insert(0, %r)
import torch
from tests.test_gpu_fullsize import _inputs, _mdcn2d
x, off, m, w, b, go = _inputs(32, 256, 256, (56, 56), 9, 2, True)
out, g, _ = _mdcn2d(x, off, m, w, b, go, "auto")
h = 16
parts = [_mdcn2d(x[s].contiguous(), off[s].contiguous(), m[s].contiguous(), w, b, go[s].contiguous(), "auto")
         for s in (slice(0, h), slice(h, None))]
assert torch.equal(torch.cat([p[0] for p in parts]), out), "output"
assert torch.equal(torch.cat([p[1][1] for p in parts]), g[1]), "grad_offset"
assert torch.equal(torch.cat([p[1][2] for p in parts]), g[2]), "grad_mask"
print("BITSTABLE_OK")
"""


def test_cfg2_shards_are_bit_identical_with_the_forward_tail_split_off():
    """MDCONV_FWD_TAIL=0 is the bit-stable switch INTEGRATION.md names: without the tap-range tail of the forward's
    last dispatch round (whose tile selection follows batch size, CU count and occupancy) the output of a batch shard
    equals the full-batch output bit for bit, and so do grad_offset / grad_mask (single owner per (tap, pixel), fixed
    K order); read once per process, hence a child."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", BITSTABLE_CODE % root], cwd=root, env=dict(os.environ, MDCONV_FWD_TAIL="0"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "BITSTABLE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


# --------------------------------------------------------------------------- cfg4 (3-D, trilinear)
def test_cfg4_deform_conv3d_mfma_equals_direct_and_conv3d():
    """DeformConv3d 3x3x3, C=64, 32^3, B=8, fp32 (BASELINE.json configs[3])."""
    from modulated_deform_conv_amd import MDCONV_CUDA as M, _capi
    x, off, _, w, b, go = _inputs(8, 64, 64, (32, 32, 32), 27, 3, False, seed=4)
    geo = (3, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 64, True)

    def run(path, offs):
        prev = _capi.set_path(path)
        try:
            out = torch.empty_like(go)
            M.deform_conv3d_forward_cuda(x, w, b, offs, out, *geo)
            gi, gw, gb, goff = torch.zeros_like(x), torch.zeros_like(w), torch.zeros_like(b), torch.zeros_like(offs)
            M.deform_conv3d_backward_cuda(x, w, b, offs, gi, gw, gb, goff, go, *geo)
            return out, gi, goff, gw, gb, _capi.last_path()
        finally:
            _capi.set_path(prev)

    a = run("mfma", off)
    d = run("direct", off)
    assert a[5] == "mfma" and d[5] == "direct"
    for name, p, q in zip(("output", "grad_input", "grad_offset", "grad_weight", "grad_bias"), a, d):
        assert_close(name, p, q, 1e-4)
    z = run("auto", torch.zeros_like(off))
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    ref = F.conv3d(xr, wr, b, 1, 1)
    ref.backward(go)
    assert_close("conv3d output", z[0], ref, 1e-4)
    assert_close("conv3d grad_input", z[1], xr.grad, 1e-4)
    assert_close("conv3d grad_weight", z[3], wr.grad, 1e-4)


# --------------------------------------------------------------------------- cfg3 / cfg5 shapes (fp16)
def test_cfg3_shape_fp16_grouped_matches_fp32():
    """ModulatedDeformConv2d C=256 56x56 group=32 deformable_group=4 fp16 (configs[2], one 4-image
    slice of a GPU's shard): fp16 storage vs the same op in fp32."""
    from modulated_deform_conv_amd import MDCONV_CUDA as M
    g = torch.Generator().manual_seed(3)
    B, C, O, G, DG, K = 4, 256, 256, 32, 4, 9
    x = torch.randn(B, C, 56, 56, generator=g).cuda()
    off = torch.randn(B, DG * 2 * K, 56, 56, generator=g).cuda()
    m = torch.sigmoid(torch.randn(B, DG * K, 56, 56, generator=g)).cuda()
    w = ((torch.rand(O, C // G, 3, 3, generator=g) * 2 - 1) / math.sqrt(C * K)).cuda()
    b = x.new_empty(0)
    go = torch.randn(B, O, 56, 56, generator=g).cuda()
    geo = (3, 3, 1, 1, 1, 1, 1, 1, G, DG, 64, False)
    h = lambda t: t.half()
    out16 = M.modulated_deform_conv2d_forward_cuda(h(x), h(w), h(b), h(off), h(m), *geo)
    out32 = M.modulated_deform_conv2d_forward_cuda(h(x).float(), h(w).float(), b, h(off).float(), h(m).float(), *geo)
    assert_close("output", out16, out32, 2e-2)
    g16 = M.modulated_deform_conv2d_backward_cuda(h(x), h(w), h(b), h(off), h(m), h(go), *geo)
    g32 = M.modulated_deform_conv2d_backward_cuda(h(x).float(), h(w).float(), b, h(off).float(), h(m).float(),
                                                 h(go).float(), *geo)
    for name, p, q in zip(("grad_input", "grad_offset", "grad_mask", "grad_weight"), g16, g32):
        assert_close(name, p, q, 3e-2)


def test_cfg5_shape_fp16_3d_dilated_forward():
    """ModulatedDeformConv3d C=128 16x64x64 dilation 2 fp16 (configs[4]), B=1 forward vs fp32."""
    from modulated_deform_conv_amd import MDCONV_CUDA as M
    g = torch.Generator().manual_seed(5)
    B, C, O, K = 1, 128, 128, 27
    sp = (16, 64, 64)
    x = torch.randn(B, C, *sp, generator=g).cuda().half()
    off = torch.randn(B, 3 * K, *sp, generator=g).cuda().half()
    m = torch.sigmoid(torch.randn(B, K, *sp, generator=g)).cuda().half()
    w = ((torch.rand(O, C, 3, 3, 3, generator=g) * 2 - 1) / math.sqrt(C * K)).cuda().half()
    b = x.new_empty(0)
    geo = (3, 3, 3, 1, 1, 1, 2, 2, 2, 2, 2, 2, 1, 1, 64, False)
    out16 = torch.empty(B, O, *sp, device="cuda", dtype=torch.float16)
    M.modulated_deform_conv3d_forward_cuda(x, w, b, off, m, out16, *geo)
    out32 = torch.empty(B, O, *sp, device="cuda")
    M.modulated_deform_conv3d_forward_cuda(x.float(), w.float(), b.float(), off.float(), m.float(), out32, *geo)
    assert_close("output", out16, out32, 2e-2)
