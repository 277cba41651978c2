"""Seeded random shapes: the MFMA path against the direct path (two independent implementations of
the same semantics) and, for the smaller half, against the CPU oracle.  Catches edge cases the
hand-picked cases miss: ragged channel counts, batch tails, odd extents, stride / dilation /
padding mixes, conv groups and deformable groups on the matrix-core backward."""
import random

import pytest
import torch

from tests.cases import D2, D3, M2, M3, _c, case_f32_wide, case_hp_wide, make_inputs
from tests.util import assert_close, run_oracle, run_product

pytestmark = pytest.mark.gpu

_PATHS = []   # (forward path, backward path) of every case run with path selection on "auto"


def _random_case(seed):
    r = random.Random(seed)
    nd = r.choice([2, 2, 2, 3])
    modulated = r.random() < 0.6
    op = {(2, False): D2, (2, True): M2, (3, False): D3, (3, True): M3}[(nd, modulated)]
    groups = r.choice([1, 1, 1, 2, 4])
    dg_choice = r.choice([1, 1, 2])
    # channels: multiples of 8 (MFMA backward), >= 16, divisible by groups and deformable groups
    if dg_choice > 1:
        cdg = r.choice([16, 32, 64, 128])
        C = cdg * dg_choice
        while C % groups:
            groups //= 2
    else:
        C = r.choice([16, 24, 32, 40, 48, 64, 72, 96, 136]) * (1 if r.random() < 0.8 else 2)
        while C % groups:
            groups = max(1, groups // 2)
    O = r.choice([16, 17, 20, 32, 33, 48, 64, 80, 130]) * groups // groups
    O = (O + groups - 1) // groups * groups
    k = r.choice([1, 2, 3, 3, 3]) if nd == 2 else r.choice([1, 2, 3])
    stride = r.choice([1, 1, 2])
    dil = r.choice([1, 1, 2])
    pad = r.choice([0, 1, dil * (k - 1) // 2 + (1 if k > 1 else 0)])
    lo = dil * (k - 1) + 1
    size = tuple(r.randint(max(lo, 3), 13 if nd == 2 else 7) for _ in range(nd))
    size = size[:-1] + (max(size[-1], 2),)
    B = r.choice([1, 2, 3])
    return _c("fuzz%d" % seed, op, B, C, O, size, k, stride=stride, padding=pad, dilation=dil,
              groups=groups, dgroups=dg_choice, in_step=r.choice([1, 2, 64]), bias=r.random() < 0.5,
              tier="medium", seed=500 + seed, offset_scale=r.choice([0.5, 1.0, 3.0]))


@pytest.mark.parametrize("seed", range(40))
def test_mfma_equals_direct_and_oracle_on_random_shapes(seed):
    case = _random_case(seed)
    t = make_inputs(case, device="cuda")
    out_a, g_a, paths = run_product(case, t, "auto")
    _PATHS.append(tuple(paths))
    out_d, g_d, _ = run_product(case, t, "direct")
    assert_close("output", out_a, out_d, 1e-4)
    for k, v in g_a.items():
        if v is not None and g_d[k] is not None:
            assert_close(k, v, g_d[k], 1e-4)
    if seed % 2 == 0:
        want_out, want = run_oracle(case, t, torch.float32)
        assert_close("output/oracle", out_a, want_out, 1e-4)
        for k, v in g_a.items():
            if v is not None and want[k] is not None:
                assert_close(k + "/oracle", v, want[k], 1e-4)


def test_fuzz_exercises_the_matrix_core_path():
    """The random shapes are drawn so that most of them qualify for the MFMA kernels."""
    if len(_PATHS) < 40:
        pytest.skip("needs the 40 random cases of this module in the same session")
    bwd = sum(1 for p in _PATHS if p[1] == "mfma")
    fwd = sum(1 for p in _PATHS if p[0] == "mfma")
    assert bwd >= 30 and fwd >= 20, (fwd, bwd)


def _random_hp_case(seed):
    """Random shapes for the native 16-bit kernels: channel counts around the 32-channel blocks (ragged
    ones are padded), conv groups, deformable groups of 32 / 64 channels, odd extents and batch tails --
    hp_bwd3 + hp_gemm2 (one group, power-of-two padded C_in), hp_bwd2, hp_bwd, the two-pass gathers."""
    r = random.Random(1000 + seed)
    nd = r.choice([2, 2, 3])
    modulated = r.random() < 0.6
    op = {(2, False): D2, (2, True): M2, (3, False): D3, (3, True): M3}[(nd, modulated)]
    dg = r.choice([1, 1, 1, 2, 4, 8])
    if dg > 1:
        C = dg * r.choice([32, 64])
        if C > 256:
            C = 256
            dg = 256 // r.choice([32, 64])
        groups = r.choice([1, 2]) if C % 2 == 0 else 1
    else:
        C = r.choice([8, 24, 32, 40, 64, 96, 128, 136, 256])
        groups = r.choice([1, 1, 1, 2, 4, 8])
        while C % groups:
            groups //= 2
    O = r.choice([8, 24, 32, 48, 64, 100, 128, 200, 256])
    O = (O + groups - 1) // groups * groups
    k = r.choice([1, 2, 3, 3]) if nd == 2 else r.choice([1, 2, 3])
    stride = r.choice([1, 1, 2])
    dil = r.choice([1, 1, 2])
    pad = r.choice([0, 1, dil * (k - 1) // 2 + (1 if k > 1 else 0)])
    lo = dil * (k - 1) + 1
    size = tuple(r.randint(max(lo, 3), 12 if nd == 2 else 6) for _ in range(nd))
    size = size[:-1] + (max(size[-1], 2),)
    return _c("hpfuzz%d" % seed, op, r.choice([1, 2, 3]), C, O, size, k, stride=stride, padding=pad, dilation=dil,
              groups=groups, dgroups=dg, in_step=64, bias=r.random() < 0.5, tier="medium", seed=1500 + seed,
              offset_scale=r.choice([0.5, 1.0, 3.0]))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("seed", range(24))
def test_16bit_random_shapes_against_the_oracle(seed, dtype):
    if dtype == torch.bfloat16 and seed % 3:
        pytest.skip("bf16: every third shape")
    case = _random_hp_case(seed)
    t = make_inputs(case, dtype=dtype, device="cuda")
    out, grads, _ = run_product(case, t, "auto")
    want_out, want = run_oracle(case, {k: (None if v is None else v.float()) for k, v in t.items()}, torch.float32)
    tol = 1e-2 if dtype == torch.float16 else 4e-2
    assert_close("output", out.float(), want_out, tol)
    for k, v in grads.items():
        if v is not None and want[k] is not None:
            assert_close(k, v.float(), want[k], tol)


@pytest.mark.parametrize("seed", range(100, 132))
def test_wide_geometry_fp32(seed):
    """Kernel extents up to 7 (2-D) / 5 (3-D) per axis -- 16 to 125 taps, where the case list stops at 27 --, strides and
    dilations up to 3, extents down to one output position, offsets of up to 8 pixels: matrix-core path against the generic
    path, every second shape also against the oracle (campaign: profiles/r05_experiments.md 25, 4134 shapes clean)."""
    case = case_f32_wide(seed)
    t = make_inputs(case, device="cuda")
    out_a, g_a, _ = run_product(case, t, "auto")
    out_d, g_d, _ = run_product(case, t, "direct")
    assert_close("output", out_a, out_d, 1e-4)
    for k, v in g_a.items():
        if v is not None and g_d[k] is not None:
            assert_close(k, v, g_d[k], 1e-4)
    if seed % 2 == 0 and case["B"] * case["C"] * case["O"] < 200000:
        want_out, want = run_oracle(case, t, torch.float32)
        assert_close("output/oracle", out_a, want_out, 1e-4)
        for k, v in g_a.items():
            if v is not None and want[k] is not None:
                assert_close(k + "/oracle", v, want[k], 1e-4)


@pytest.mark.parametrize("seed", range(100, 124))
def test_wide_geometry_16bit(seed):
    """The same geometry on the native 16-bit kernels against the oracle (fp16; every third shape bf16)."""
    dtype = torch.bfloat16 if seed % 3 == 0 else torch.float16
    case = case_hp_wide(seed)
    t = make_inputs(case, dtype=dtype, device="cuda")
    out, grads, _ = run_product(case, t, "auto")
    want_out, want = run_oracle(case, {k: (None if v is None else v.float()) for k, v in t.items()}, torch.float32)
    tol = 1e-2 if dtype == torch.float16 else 4e-2
    assert_close("output", out.float(), want_out, tol)
    for k, v in grads.items():
        if v is not None and want[k] is not None:
            assert_close(k, v.float(), want[k], tol)


# The eleven per-element misses of the round-5 wide campaigns (profiles/r05_experiments.md 25, 30): 3-D shapes of 64-125 taps
# (and one 3 x 2 input) whose output positions mostly sample padding, so that the criterion's rms is tiny; every one within
# 2.3x of the campaign's tolerance, on grad_offset / grad_mask (one grad_weight).
KNOWN_WIDE_MISSES = [(561, torch.bfloat16), (1360, torch.float16), (1374, torch.bfloat16), (1653, torch.bfloat16),
                     (2542, torch.float16), (5808, torch.bfloat16), (10161, torch.bfloat16), (10788, torch.bfloat16),
                     (30340, torch.float16), (30846, torch.bfloat16), (32169, torch.bfloat16)]


@pytest.mark.parametrize("seed, dtype", KNOWN_WIDE_MISSES, ids=lambda v: str(v).replace("torch.", ""))
def test_wide_16bit_misses_are_the_rounding_of_the_reference_s_own_half_buffers(seed, dtype):
    """VERDICT r5 "what's weak" 2.  The native 16-bit kernels round grad_col between GEMM-1 and the coordinate sums, and the
    column values before GEMM-2 -- exactly where the REFERENCE rounds with half tensors: its `grad_columns` and `columns`
    buffers are tensors of the input's type (mdeformable_conv.cu:396-397; 3-D mdeformable_conv3d.cu:494-497).  Against the
    oracle with those two buffers stored in the tensors' type (everything else fp32) the eleven shapes pass at the campaign's
    STANDARD tolerance; against the all-fp32 oracle they stay within 2.5x of it, which pins the size of the deviation so
    that a regression in these kernels shows up here."""
    case = case_hp_wide(seed)
    t = make_inputs(case, dtype=dtype, device="cuda")
    out, grads, _ = run_product(case, t, "auto")
    f32 = {k: (None if v is None else v.float()) for k, v in t.items()}
    tol = 1e-2 if dtype == torch.float16 else 4e-2
    want_out, want_ref = run_oracle(case, f32, torch.float32, intermediates=dtype)
    assert_close("output", out.float(), want_out, tol)
    for k, v in grads.items():
        if v is not None and want_ref[k] is not None:
            assert_close(k + " (oracle with the reference's 16-bit buffers)", v.float(), want_ref[k], tol)
    _, want32 = run_oracle(case, f32, torch.float32)
    for k, v in grads.items():
        if v is not None and want32[k] is not None:
            assert_close(k + " (all-fp32 oracle)", v.float(), want32[k], tol, 2.5 * tol)


def test_wide_borderline_miss_of_the_final_round6_campaign():
    """Seed 310540 of the last `fuzz_more.py --wide` campaign of round 6 (profiles/r06_experiments.md 30): MDCN3d 64 -> 128, 125
    taps, 4 deformable groups, one output position per image -- ONE grad_weight element of 1 M (magnitude 0.018 where the tensor's
    rms is 0.36: a cancelling sum of 16-bit column values) at 1.08x the per-element tolerance against the oracle with the
    reference's 16-bit buffers.  tools/fuzz_repro.py classes it as rounding (bf16: 2.8x; through the fp32 kernels: 4e-4).  Pinned
    at the standard scaled tolerance and 1.5x the per-element one against both oracles, so that a regression shows up here."""
    case = case_hp_wide(310540)
    t = make_inputs(case, dtype=torch.float16, device="cuda")
    out, grads, _ = run_product(case, t, "auto")
    f32 = {k: (None if v is None else v.float()) for k, v in t.items()}
    tol = 1e-2
    want_out, want_ref = run_oracle(case, f32, torch.float32, intermediates=torch.float16)
    _, want32 = run_oracle(case, f32, torch.float32)
    assert_close("output", out.float(), want_out, tol)
    for k, v in grads.items():
        if v is not None and want_ref[k] is not None:
            assert_close(k + " (oracle with the reference's 16-bit buffers)", v.float(), want_ref[k], tol, 1.5 * tol)
            assert_close(k + " (all-fp32 oracle)", v.float(), want32[k], tol, 1.5 * tol)

