"""GPU parity of the NATIVE 16-bit kernels (hp_*.hip: fp16 / bf16 operands straight into
v_mfma_f32_32x32x16_{f16,bf16}, fp32 coordinates / weights / accumulators) against the fp32 oracle
run on the same 16-bit-rounded inputs.  Reference dtype dispatch: mdeformable_conv.cu:101, 334
(AT_DISPATCH_FLOATING_TYPES_AND_HALF); bf16 is the SURVEY.md section 8f-3 extension.

Tolerance: 5e-3 (fp16) / 3e-2 (bf16) on both criteria of tests.util.assert_close -- the outputs
themselves are rounded to 11 / 8 significant bits (measured worst case 1.9e-3 / 1.4e-2).
"""
import os

import pytest
import torch

from tests.cases import CASES, CASE_BY_NAME, make_inputs, _c, M2, M3, D2, D3
from tests.util import assert_close, run_oracle, run_product

pytestmark = pytest.mark.gpu

TOL = {torch.float16: 5e-3, torch.bfloat16: 3e-2}

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
    # more than 4 deformable groups: beyond the (pixel, deformable group) state table of the
    # line-wide backward, i.e. the lane = pixel kernel (hp_bwd.hip) is what runs
    _c("hp_mdcn2d_c256_o64_dg8", M2, 2, 256, 64, (9, 8), 3, dgroups=8, seed=117),
    _c("hp_dcn3d_c256_o32_dg8", D3, 1, 256, 32, (4, 5, 6), 3, dgroups=8, bias=False, seed=118),
    # 3-D with rows of 32 / 64 pixels and 8 | rows: the blocked tile order of the pixel-stationary kernels
    _c("hp_mdcn3d_c32_o32_rows32", M3, 2, 32, 32, (3, 16, 32), 3, seed=119),
    _c("hp_dcn3d_c64_o32_rows64_dil2", D3, 1, 64, 32, (4, 8, 64), 3, padding=2, dilation=2, bias=False, seed=120),
    # pixel-stationary backward with 16 k-steps of output channels (C_out > 128: the register-heaviest variant)
    _c("hp_mdcn2d_c64_o256", M2, 1, 64, 256, (9, 8), 3, seed=121),
    _c("hp_dcn3d_c32_o200", D3, 1, 32, 200, (3, 5, 6), 3, bias=False, seed=122),
    # W^T slab of one tap beyond 48 KB (round 6): the pixel-stationary backward reads its A fragments straight from global
    # memory (hp_bwd3.hip, kWG) -- 256 x >= 128 and 128 x 256 channels, 2-D and 3-D, tiles that straddle images, 25 taps
    _c("hp_dcn3d_c256_o128_wg", D3, 1, 256, 128, (3, 5, 6), 3, seed=123),
    _c("hp_mdcn3d_c128_o256_wg", M3, 1, 128, 256, (4, 5, 5), 3, bias=False, seed=124),
    _c("hp_mdcn2d_c128_o200_wg_b3", M2, 3, 128, 200, (7, 9), 3, seed=125),
    _c("hp_dcn2d_c256_o256_wg_k5", D2, 2, 256, 256, (11, 10), 5, padding=2, seed=126),
    # deformable groups on the pixel-stationary backward (round 6): a (tap, group) pair is a gather unit; groups of 16, 32, 64
    # and 128 channels, 2 and 4 groups, with a staged W^T slab and with A fragments from global memory, 2-D and 3-D
    _c("hp_mdcn2d_c64_dg4_o64", M2, 2, 64, 64, (9, 10), 3, dgroups=4, seed=127),
    _c("hp_dcn2d_c32_dg2_o48", D2, 3, 32, 48, (7, 9), 3, dgroups=2, bias=False, seed=128),
    _c("hp_dcn2d_c128_dg4_o96_s2", D2, 2, 128, 96, (12, 11), 3, stride=2, dgroups=4, seed=129),
    _c("hp_mdcn2d_c256_dg4_o256", M2, 1, 256, 256, (8, 9), 3, dgroups=4, seed=130),
    _c("hp_mdcn2d_c256_dg2_o128_dil2", M2, 2, 256, 128, (9, 8), 3, padding=2, dilation=2, dgroups=2, bias=False, seed=141),
    _c("hp_mdcn3d_c64_dg2_o64", M3, 1, 64, 64, (4, 5, 6), 3, dgroups=2, seed=142),
    _c("hp_dcn3d_c128_dg4_o64", D3, 1, 128, 64, (3, 6, 5), 3, dgroups=4, bias=False, seed=143),
    _c("hp_mdcn3d_c64_dg4_o32_big_offsets", M3, 2, 64, 32, (4, 4, 5), 3, dgroups=4, seed=144, offset_scale=3.0),
    # group-padded layout (round 6, hp_host.hip group_padded): deformable groups of 24 / 48 / 80 / 40 / 12 channels run as groups
    # of 32 / 64 / 128 / 64 / 16 with the padding channels zero; 2, 3 and 4 groups, 2-D and 3-D, with and without mask
    _c("hp_mdcn2d_c96_dg4_o96_pad", M2, 2, 96, 96, (9, 10), 3, dgroups=4, seed=145),
    _c("hp_dcn2d_c192_dg4_o64_pad", D2, 1, 192, 64, (8, 9), 3, dgroups=4, bias=False, seed=146),
    _c("hp_mdcn2d_c160_dg2_o48_s2_pad", M2, 2, 160, 48, (12, 11), 3, stride=2, dgroups=2, seed=147),
    _c("hp_mdcn2d_c72_dg3_o40_pad", M2, 2, 72, 40, (7, 9), 3, dgroups=3, seed=148),
    _c("hp_mdcn3d_c48_dg2_o32_pad", M3, 1, 48, 32, (4, 5, 6), 3, dgroups=2, bias=False, seed=149),
    _c("hp_dcn3d_c80_dg2_o64_dil2_pad", D3, 1, 80, 64, (4, 6, 5), 3, padding=2, dilation=2, dgroups=2, seed=150),
    _c("hp_mdcn2d_c48_dg4_o64_big_offsets_pad", M2, 3, 48, 64, (8, 7), 3, dgroups=4, seed=151, offset_scale=4.0),
    # one deformable group, 96 / 160 / 224 padded channels: hp_bwd3 runs them padded to 128 / 256 (hp_host.hip width_padded) where its
    # size rule applies -- here only in the forced child (tests/test_gpu_hp_forced.py, MDCONV_HP_BWD=4); the tap-stationary kernels by default
    _c("hp_mdcn2d_c96_o64_wpad", M2, 2, 96, 64, (9, 10), 3, seed=152),
    _c("hp_dcn3d_c160_o48_wpad", D3, 1, 160, 48, (4, 5, 6), 3, bias=False, seed=153),
    _c("hp_mdcn2d_c200_o40_s2_wpad", M2, 2, 200, 40, (12, 11), 3, stride=2, seed=154),
]

CASE_BY_HP = {c["name"]: c for c in HP_CASES}

# 16-bit shapes the native kernels reject in at least one direction (hp_supported): more than 256
# input channels / a block's output range above 256 in the backward, deformable groups whose padded
# form (group_padded, hp_host.hip) exceeds 256 channels.  fp16 AND bf16 must still work (fp32 copies through the fp32 kernels, else the
# shape-generic kernels) -- ADVICE round 2: bf16 used to end in "unknown dtype 3".
FALLBACK_CASES = [
    _c("fb_mdcn2d_c512_o64", M2, 1, 512, 64, (7, 6), 3, seed=131),
    _c("fb_mdcn2d_c320_o64_dg4", M2, 2, 320, 64, (8, 7), 3, dgroups=4, seed=132),   # groups of 80: padded to 128 = 512 channels
    _c("fb_dcn3d_c24_o8_g2", D3, 1, 24, 8, (4, 5, 4), 3, groups=2, bias=False, seed=133),
]


def _check(case, dtype, expect_hp=True, tol=None):
    from modulated_deform_conv_amd import _capi
    t = make_inputs(case, dtype=dtype, device="cuda")
    out, grads, paths = run_product(case, t, "auto")
    torch.cuda.synchronize()
    # (with the older backward kernels forced -- tests/test_gpu_hp_forced.py -- deformable groups of 16 channels have no
    # native backward: only the pixel-stationary kernel takes them; the case still runs, on the fp32 kernels)
    cdg = case["C"] // case["dgroups"]
    if case["name"].endswith("_pad"):   # the group size the kernels see
        cdg = 1 << (cdg - 1).bit_length() if case["dgroups"] in (2, 4) else (cdg + 31) // 32 * 32
    forced_old = os.environ.get("MDCONV_HP_BWD") in ("1", "2") and case["dgroups"] > 1 and cdg % 32
    if expect_hp and not forced_old:
        assert _capi.last_kernels() == "hp", (_capi.last_kernels(), paths)
    want_out, want = run_oracle(case, {k: (None if v is None else v.float()) for k, v in t.items()}, torch.float32)
    tol = tol or TOL[dtype]
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


def test_wide_backward_runs_on_the_pixel_stationary_kernel():
    """256 -> 256 channels (the fp16 twin of the headline shape) used to run its backward on hp_bwd2, whose instance with 16
    k-steps spills 744 bytes per lane; since round 6 hp_bwd3 takes it (A fragments from global memory) as soon as there is
    more work than a few (tile, tap) pairs per CU -- with 4 deformable groups as well."""
    from modulated_deform_conv_amd import _capi
    for dg in (1, 4):
        case = _c("hp_wide_mdcn2d_c256_o256_28_dg%d" % dg, M2, 8, 256, 256, (28, 28), 3, dgroups=dg, seed=170 + dg)   # 49 tiles x 9 taps
        t = make_inputs(case, dtype=torch.float16, device="cuda")
        _capi.profile_enable(True)
        _capi.profile_reset()
        try:
            out, grads, _ = run_product(case, t, "auto")
            torch.cuda.synchronize()
            names = set(_capi.profile_read())
        finally:
            _capi.profile_enable(False)
        assert "hp_bwd3_kernel" in names and "hp_gemm2_kernel" in names, names
        want_out, want = run_oracle(case, {k: (None if v is None else v.float()) for k, v in t.items()}, torch.float32)
        assert_close("output", out.float(), want_out, TOL[torch.float16])
        for key, g in grads.items():
            if want[key] is not None:
                assert_close(key, g.float(), want[key], TOL[torch.float16])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("case", FALLBACK_CASES, ids=lambda c: c["name"])
def test_16bit_shapes_outside_the_native_kernels(case, dtype):
    _check(case, dtype, expect_hp=False)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_autocast_on_a_shape_the_native_backward_rejects(dtype):
    """torch.autocast on a C_in = 512 layer: the forward qualifies for the native 16-bit kernels,
    the backward does not (more than 8 channel blocks) -- loss.backward() must still work."""
    from modulated_deform_conv_amd.modulated_deform_conv import ModulatedDeformConv2d
    case = FALLBACK_CASES[0]
    t = make_inputs(case, dtype=torch.float32, device="cuda")
    mod = ModulatedDeformConv2d(512, 64, 3, padding=1, bias=True).cuda()
    with torch.no_grad():
        mod.weight.copy_(t["weight"]); mod.bias.copy_(t["bias"])
    x, off, m = (t[k].clone().requires_grad_() for k in ("input", "offset", "mask"))
    with torch.autocast("cuda", dtype=dtype):
        out = mod(x, off, m)
    assert out.dtype == dtype
    out.backward(t["grad_output"].to(dtype))
    r = lambda v: v.to(dtype).float()
    want_out, want = run_oracle(case, {k: (None if v is None else r(v)) for k, v in t.items()}, torch.float32)
    tol = TOL[dtype]
    assert_close("output", out.float(), want_out, tol)
    assert x.grad.dtype == torch.float32 and mod.weight.grad.dtype == torch.float32
    assert_close("grad_input", x.grad, want["grad_input"], tol)
    assert_close("grad_offset", off.grad, want["grad_offset"], tol)
    assert_close("grad_mask", m.grad, want["grad_mask"], tol)
    assert_close("grad_weight", mod.weight.grad, want["grad_weight"], tol)
    assert_close("grad_bias", mod.bias.grad, want["grad_bias"], tol)


def test_channels_last_input_on_a_shape_the_native_backward_rejects():
    """A channels_last fp16 input with C_in = 512: consumed in place by the forward, and the
    backward of the same autograd Function (which saved the channels-last tensor) falls back to a
    contiguous copy instead of raising."""
    from modulated_deform_conv_amd.modulated_deform_conv import modulated_deform_conv2d
    case = FALLBACK_CASES[0]
    t = make_inputs(case, dtype=torch.float16, device="cuda")
    res = []
    for fmt in (torch.contiguous_format, torch.channels_last):
        x = t["input"].detach().clone(memory_format=fmt).requires_grad_()
        off, m, w = (t[k].clone().requires_grad_() for k in ("offset", "mask", "weight"))
        out = modulated_deform_conv2d(x, off, m, w, t["bias"], 1, 1, 1, 1, 1, 64)
        out.backward(t["grad_output"])
        res.append((out, x.grad, off.grad, m.grad, w.grad))
    # (round 5: this few-tile, 72-stage forward runs on the fp32 kernels for the contiguous input and on the native kernels
    # for the channels-last one -- hp_forward_preferred -- so the two outputs agree to fp16 rounding, not bit for bit)
    assert_close("output", res[1][0].float(), res[0][0].float(), 2e-3, 1e-2)
    for a, b, name in zip(res[0][1:], res[1][1:], ("grad_input", "grad_offset", "grad_mask", "grad_weight")):
        assert_close(name, b.float(), a.float(), 2e-3, 1e-2)


@pytest.mark.parametrize("case", [c for c in CASES if c["tier"] == "medium"], ids=lambda c: c["name"])
def test_medium_cases_fp16(case):
    """Every medium case of the shared list in fp16, whichever kernels the dispatcher picks.  1e-2: the
    worst case of this list (stride-2 3-D, grad_input: sums of fp16-rounded grad_col rows with
    cancellation) measures 7.3e-3 on the per-element criterion."""
    _check(case, torch.float16, expect_hp=False, tol=1e-2)


@pytest.mark.parametrize("dtype, scale", [(torch.bfloat16, 1.0e5), (torch.bfloat16, 1.0e-7), (torch.float16, 100.0),
                                          (torch.float16, 1.0e-2)], ids=["bf16_1e5", "bf16_1e-7", "fp16_1e2", "fp16_1e-2"])
@pytest.mark.parametrize("name", ["hp_mdcn2d_c256_o256_g32_dg4", "hp_mdcn2d_c64_o96_s2"])
def test_2d_scatter_entries_carry_the_masks_range(name, dtype, scale):
    """The 2-D scatter-list entries hold bilinear weight x MASK products as packed 16-bit pairs (hp_col2im.hip).  They
    are packed in the tensors' own type, so a bf16 mask of 1e5 (above the fp16 range) or 1e-7 (products below the
    fp16 subnormals) scales grad_input like any other mask (advisor, round 4); fp16 masks inside the fp16 range too."""
    from modulated_deform_conv_amd import _capi
    case = CASE_BY_HP[name]
    t = make_inputs(case, dtype=dtype, device="cuda")
    t["mask"] = (t["mask"].float() * scale).to(dtype)
    out, grads, _ = run_product(case, t, "auto")
    torch.cuda.synchronize()
    assert _capi.last_kernels() == "hp"
    want_out, want = run_oracle(case, {k: (None if v is None else v.float()) for k, v in t.items()}, torch.float32)
    assert_close("output", out.float(), want_out, TOL[dtype])
    for key in ("grad_input", "grad_offset", "grad_weight"):
        assert_close(key, grads[key].float(), want[key], TOL[dtype])
    # grad_mask does not scale with the mask: compare it relative to the (unscaled) oracle as usual
    assert_close("grad_mask", grads["grad_mask"].float(), want["grad_mask"], TOL[dtype])


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


def test_backward_kernel_choice_follows_shape_and_size():
    """Which 16-bit backward kernel runs (seen through the profile slot names).  One conv group and 1 / 2 / 4 deformable groups:
    the pixel-stationary hp_bwd3 once the grid fills the chip (more than one 128-pixel tile per CU) or where hp_bwd2 cannot
    take the shape (16-channel groups) or would spill (16 k-steps with more than a few taps x tiles); the tap-stationary hp_bwd2
    for small grids (parallel over the taps) and for conv groups; hp_bwd beyond four deformable groups."""
    from modulated_deform_conv_amd import _capi
    by = {c["name"]: c for c in HP_CASES}
    big = _c("hp_choice_mdcn2d_c64_o64_b12_56", M2, 12, 64, 64, (56, 56), 3, seed=160)        # 294 tiles of 128 pixels
    wide = _c("hp_choice_mdcn2d_c64_o256_b4_56", M2, 4, 64, 256, (56, 56), 3, seed=161)       # 16 k-steps, 98 tiles x 9 taps
    # 16 k-steps in 3-D, 256 channels: hp_bwd2 (spilling instance) up to ~36 tiles, hp_bwd3 beyond (experiment log 16)
    few3 = _c("hp_choice_mdcn3d_c256_o256_b2", M3, 2, 256, 256, (4, 14, 14), 3, seed=162)       # 13 tiles x 27 taps
    many3 = _c("hp_choice_mdcn3d_c256_o256_b8", M3, 8, 256, 256, (4, 14, 14), 3, seed=163)      # 49 tiles
    wpad = _c("hp_choice_mdcn2d_c96_o32_b12_56", M2, 12, 96, 32, (56, 56), 3, seed=164)        # 96 channels run as 128: 294 tiles
    for case, want in ((big, "hp_bwd3_kernel"), (wide, "hp_bwd3_kernel"), (by["hp_mdcn2d_c64_dg4_o64"], "hp_bwd3_kernel"),
                       (few3, "hp_bwd2_kernel"), (many3, "hp_bwd3_kernel"), (wpad, "hp_bwd3_kernel"),
                       (by["hp_mdcn3d_c128_o128_dil2"], "hp_bwd2_kernel"), (by["hp_mdcn2d_c64_o256"], "hp_bwd2_kernel"),
                       (by["hp_mdcn2d_c256_o256_g32_dg4"], "hp_bwd2_kernel"), (by["hp_mdcn2d_c256_o64_dg8"], "hp_bwd_kernel")):
        t = make_inputs(case, dtype=torch.float16, device="cuda")
        _capi.profile_enable(True)
        _capi.profile_reset()
        run_product(case, t, "auto")
        torch.cuda.synchronize()
        _capi.profile_enable(False)
        assert want in _capi.profile_read(), (case["name"], _capi.profile_read())


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


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_few_tile_many_stage_forward_runs_on_the_fp32_kernels(dtype):
    """A 16-bit forward of a few 128-pixel tiles over many K stages (C_in = 512 at 7 x 7: one tile, 72 stages) takes one whole
    tile time on the native kernel; the library runs it on the fp32 matrix kernels through fp32 copies instead
    (hp_forward_preferred, hp_host.hip) -- same oracle tolerance -- while the backward and a channels-last input (which only
    the native kernels read in place) stay on the native kernels."""
    from modulated_deform_conv_amd import MDCONV_CUDA as M, _capi
    case = _c("hp_route_c512_7x7", M2, 2, 512, 64, (7, 7), 3, seed=131)
    t = make_inputs(case, dtype=dtype, device="cuda")
    geo = (3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 64, True)
    x, w, b, off, m = t["input"], t["weight"], t["bias"], t["offset"], t["mask"]
    out = M.modulated_deform_conv2d_forward_cuda(x, w, b, off, m, *geo)
    torch.cuda.synchronize()
    assert _capi.last_kernels() == "f32", _capi.last_kernels()
    want_out, want = run_oracle(case, {k: (None if v is None else v.float()) for k, v in t.items()}, torch.float32)
    assert_close("output", out.float(), want_out, TOL[dtype])
    grads = M.modulated_deform_conv2d_backward_cuda(x, w, b, off, m, t["grad_output"], *geo)
    torch.cuda.synchronize()
    assert _capi.last_kernels() == "f32" or _capi.last_kernels() == "hp"   # (C_in > 256: the 16-bit backward takes fp32 copies too)
    for key, g in zip(("grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias"), grads):
        assert_close(key, g.float(), want[key], TOL[dtype])
    out_cl = M.modulated_deform_conv2d_forward_cuda(x.contiguous(memory_format=torch.channels_last), w, b, off, m, *geo)
    torch.cuda.synchronize()
    assert _capi.last_kernels() == "hp", _capi.last_kernels()
    assert_close("output (channels-last input)", out_cl.float(), want_out, TOL[dtype])
    # the same shape takes two numeric routes depending only on the memory format of `input` (fp32 kernels on widened copies
    # vs the native 16-bit kernel, INTEGRATION.md "Routing by memory format"): both within the 16-bit tolerance of each other
    assert_close("output, contiguous vs channels-last input", out_cl.float(), out.float(), TOL[dtype])


@pytest.mark.parametrize("name,op,B,C,O,sz", [
    ("hp_route_c512_o512_7x7", M2, 8, 512, 512, (7, 7)),        # 4 tiles x 16 output blocks: 64 single-block rows, still "few"
    ("hp_route_c1024_o256_7x7_dg4", M2, 4, 1024, 256, (7, 7)),
    ("hp_route_3d_c256_o256_4x7x7", M3, 4, 256, 256, (4, 7, 7)),
], ids=lambda v: v if isinstance(v, str) else None)
def test_few_tile_rule_counts_work_not_workgroups(name, op, B, C, O, sz):
    """The few-tile rule must not depend on how many output blocks a workgroup row of the native forward holds: small grids
    run single-block rows (round 6), which multiplied the workgroup count the rule looked at and sent these forwards back to
    one-latency-chain native workgroups (2.4x slower than round 5; profiles/r06_experiments.md 15)."""
    from modulated_deform_conv_amd import MDCONV_CUDA as M, _capi
    nd = len(sz)
    case = _c(name, op, B, C, O, sz, 3, dgroups=4 if name.endswith("dg4") else 1, seed=137)
    t = make_inputs(case, dtype=torch.float16, device="cuda")
    geo = (3,) * nd + (1,) * (3 * nd) + (1, case["dgroups"], 64, True)
    x, w, b, off, m = t["input"], t["weight"], t["bias"], t["offset"], t["mask"]
    if nd == 2:
        out = M.modulated_deform_conv2d_forward_cuda(x, w, b, off, m, *geo)
    else:
        out = torch.empty_like(t["grad_output"])
        M.modulated_deform_conv3d_forward_cuda(x, w, b, off, m, out, *geo)
    torch.cuda.synchronize()
    assert _capi.last_kernels() == "f32", _capi.last_kernels()
    want_out, _ = run_oracle(case, {k: (None if v is None else v.float()) for k, v in t.items()}, torch.float32)
    assert_close("output", out.float(), want_out, TOL[torch.float16])
