"""GPU parity of the NATIVE 16-bit kernels (hp_*.hip: fp16 / bf16 operands straight into
v_mfma_f32_32x32x16_{f16,bf16}, fp32 coordinates / weights / accumulators) against the fp32 oracle
run on the same 16-bit-rounded inputs.  Reference dtype dispatch: mdeformable_conv.cu:101, 334
(AT_DISPATCH_FLOATING_TYPES_AND_HALF); bf16 is the SURVEY.md section 8f-3 extension.

Tolerance: 2e-2 (fp16) / 6e-2 (bf16) on both criteria of tests.util.assert_close -- the outputs
themselves are rounded to 11 / 8 significant bits.
"""
import pytest
import torch

from tests.cases import CASES, CASE_BY_NAME, make_inputs, _c, M2, M3, D2, D3
from tests.util import assert_close, run_oracle, run_product

pytestmark = pytest.mark.gpu

TOL = {torch.float16: 2e-2, torch.bfloat16: 6e-2}

# shapes written for this path: channel counts around the 32-channel blocks, conv groups that
# share a block (cfg3: 8 channels per group), deformable groups of 32 / 64 channels, C_out spans
HP_CASES = [
    _c("hp_mdcn2d_c32_o32", M2, 2, 32, 32, (9, 10), 3, seed=101),
    _c("hp_dcn2d_c40_o24_ragged", D2, 2, 40, 24, (7, 9), 3, bias=False, seed=102),
    _c("hp_mdcn2d_c64_o96_s2", M2, 2, 64, 96, (12, 11), 3, stride=2, seed=103),
    _c("hp_mdcn2d_c128_o160_dil2", M2, 1, 128, 160, (9, 9), 3, padding=2, dilation=2, seed=104),
    _c("hp_mdcn2d_c256_o256_g32_dg4", M2, 2, 256, 256, (10, 12), 3, groups=32, dgroups=4, bias=False, seed=105),
    _c("hp_mdcn2d_c128_o64_g4_dg2", M2, 2, 128, 64, (8, 9), 3, groups=4, dgroups=2, seed=106),
    _c("hp_dcn2d_c96_o48_g3", D2, 2, 96, 48, (8, 8), 3, groups=3, seed=107),
    _c("hp_mdcn2d_c64_dg2_big_offsets", M2, 2, 64, 32, (9, 9), 3, dgroups=2, seed=108, offset_scale=4.0),
    _c("hp_mdcn2d_k1", M2, 2, 64, 64, (7, 8), 1, padding=0, seed=109),
    _c("hp_dcn2d_k5_s2", D2, 1, 32, 64, (15, 13), 5, stride=2, padding=2, seed=110),
    _c("hp_mdcn3d_c32_o32", M3, 1, 32, 32, (5, 6, 5), 3, seed=111),
    _c("hp_mdcn3d_c128_o128_dil2", M3, 1, 128, 128, (4, 8, 8), 3, padding=2, dilation=2, bias=False, seed=112),
    _c("hp_dcn3d_c64_o32_s2", D3, 2, 64, 32, (5, 6, 7), 3, stride=2, seed=113),
    _c("hp_mdcn3d_c64_g2_dg2", M3, 1, 64, 64, (4, 5, 6), 3, groups=2, dgroups=2, seed=114),
    _c("hp_mdcn2d_c256_o256", M2, 1, 256, 256, (8, 8), 3, seed=115),
    _c("hp_mdcn2d_pixels_not_mult8", M2, 3, 32, 32, (7, 7), 3, seed=116),
]


def _check(case, dtype, expect_hp=True):
    from modulated_deform_conv_amd import _capi
    t = make_inputs(case, dtype=dtype, device="cuda")
    out, grads, paths = run_product(case, t, "auto")
    torch.cuda.synchronize()
    if expect_hp:
        assert _capi.last_kernels() == "hp", (_capi.last_kernels(), paths)
    want_out, want = run_oracle(case, {k: (None if v is None else v.float()) for k, v in t.items()}, torch.float32)
    tol = TOL[dtype]
    assert_close("output", out.float(), want_out, tol)
    for key, g in grads.items():
        if want[key] is None:
            continue
        assert_close(key, g.float(), want[key], tol)


@pytest.mark.parametrize("case", HP_CASES, ids=lambda c: c["name"])
def test_hp_fp16(case):
    _check(case, torch.float16)


@pytest.mark.parametrize("case", HP_CASES[::2], ids=lambda c: c["name"])
def test_hp_bf16(case):
    _check(case, torch.bfloat16)


@pytest.mark.parametrize("case", [c for c in CASES if c["tier"] == "medium"], ids=lambda c: c["name"])
def test_medium_cases_fp16(case):
    """Every medium case of the shared list in fp16, whichever kernels the dispatcher picks."""
    _check(case, torch.float16, expect_hp=False)


def test_non_finite_border_pixel_is_not_read():
    """A corner outside the image is never loaded (reference: `if (h_low >= 0 ...)`,
    mdeformable_conv.cu:9-34), so an Inf in a border pixel only reaches the samples that touch it."""
    case = _c("hp_inf_border", M2, 1, 32, 32, (8, 8), 3, seed=120)
    t = make_inputs(case, dtype=torch.float16, device="cuda")
    t["offset"].zero_()
    t["input"][0, :, 0, 0] = float("inf")
    out, _, _ = run_product(case, t, "auto")
    # pixel (0,0) is inside the 3x3 window (zero offsets, pad 1) of output pixels (0..1, 0..1) only
    bad = ~torch.isfinite(out[0, 0])
    assert bad[:2, :2].all() and bad.sum().item() == 4


def test_hp_accumulate_and_overwrite():
    """Caller-allocated backward: accumulate (default) adds to the buffers, overwrite mode writes
    every element (NaN-filled buffers)."""
    from modulated_deform_conv_amd import MDCONV_CUDA as M, _capi
    case = CASE_BY_NAME["cl_mdcn3d_g2_c128_o64_k2"]
    t = make_inputs(case, dtype=torch.float16, device="cuda")
    _, want, _ = run_product(case, t, "auto")
    assert _capi.last_kernels() == "hp"
    x, w, off, m, go, b = t["input"], t["weight"], t["offset"], t["mask"], t["grad_output"], t["bias"]
    geo = (2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 64, True)
    nan = lambda r: torch.full_like(r, float("nan"))
    gi, gw, gb, goff, gm = nan(x), nan(w), nan(b), nan(off), nan(m)
    with _capi.overwrite_grads():
        M.modulated_deform_conv3d_backward_cuda(x, w, b, off, m, gi, gw, gb, goff, gm, go, *geo)
    for k, g in (("grad_input", gi), ("grad_weight", gw), ("grad_bias", gb), ("grad_offset", goff), ("grad_mask", gm)):
        assert_close(k, g.float(), want[k].float(), 2e-3)
    one = lambda r: torch.ones_like(r)
    gi, gw, gb, goff, gm = one(x), one(w), one(b), one(off), one(m)
    M.modulated_deform_conv3d_backward_cuda(x, w, b, off, m, gi, gw, gb, goff, gm, go, *geo)
    for k, g in (("grad_input", gi), ("grad_weight", gw), ("grad_bias", gb), ("grad_offset", goff), ("grad_mask", gm)):
        assert_close(k, g.float() - 1, want[k].float(), 2e-2)


@pytest.mark.parametrize("name", ["hp_mdcn2d_c64_o96_s2", "hp_mdcn3d_c128_o128_dil2", "hp_mdcn2d_c256_o256_g32_dg4"])
def test_channels_last_input_is_consumed_in_place(name):
    """A channels_last / channels_last_3d fp16 input goes to the kernels as it is (no layout pass)
    and gives the results of the contiguous input; shapes the 16-bit kernels do not take keep the
    reference's "has to be contiguous" error."""
    from modulated_deform_conv_amd import MDCONV_CUDA as M
    case = {c["name"]: c for c in HP_CASES}[name]
    t = make_inputs(case, dtype=torch.float16, device="cuda")
    out, grads, _ = run_product(case, t, "auto")
    fmt = torch.channels_last if t["input"].dim() == 4 else torch.channels_last_3d
    t2 = dict(t, input=t["input"].contiguous(memory_format=fmt))
    assert not t2["input"].is_contiguous()
    out2, grads2, _ = run_product(case, t2, "auto")
    assert torch.equal(out, out2)
    for k in grads:
        if grads[k] is not None:
            assert_close(k, grads2[k].float(), grads[k].float(), 2e-3, 1e-2)
    x32 = torch.randn(2, 32, 6, 6, device="cuda").contiguous(memory_format=torch.channels_last)
    with pytest.raises(RuntimeError, match="contiguous"):
        M.modulated_deform_conv2d_forward_cuda(x32, torch.randn(8, 32, 3, 3, device="cuda"), x32.new_empty(0),
                                               torch.zeros(2, 18, 6, 6, device="cuda"), torch.ones(2, 9, 6, 6, device="cuda"),
                                               3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 64, False)
