"""GPU parity tests (``-m gpu``): the HIP path, called through the MDCONV_CUDA surface (ctypes ->
C ABI -> kernels), against the CPU oracle on the same seeded inputs.

Tolerance (BASELINE.json north_star: "within 1e-4 fp32"), read relative to scale as SURVEY.md
section 8c prescribes:  max|got - want| <= 1e-4 * max(1, max|want|)   for fp32,
1e-10 for fp64, and 5e-3 for fp16 storage (fp32 arithmetic inside; compared with the fp32 oracle
run on the fp16-rounded inputs).
"""
import pytest
import torch

from tests.cases import CASES, CASE_BY_NAME, make_inputs
from tests.util import assert_close, run_oracle, run_product

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 1e-4, torch.float64: 1e-10, torch.float16: 5e-3}


def _check(case, dtype, path):
    t = make_inputs(case, dtype=dtype, device="cuda")
    out, grads, paths = run_product(case, t, path)
    torch.cuda.synchronize()
    odt = torch.float64 if dtype == torch.float64 else torch.float32
    want_out, want = run_oracle(case, t, odt)
    tol = TOL[dtype]
    assert_close("output", out, want_out, tol)
    for key, g in grads.items():
        if want[key] is None:
            continue
        assert_close(key, g, want[key], tol)
    return paths


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_fp32_direct_path(case):
    assert _check(case, torch.float32, "direct") == ["direct", "direct"]


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_fp32_auto_path(case):
    _check(case, torch.float32, "auto")


@pytest.mark.parametrize("case", [c for c in CASES if c["name"].startswith(("mfma_", "cfg2s_", "cfg3s_mdcn2d_c256",
                                                                           "cfg4s_", "cfg5s_"))],
                         ids=lambda c: c["name"])
def test_backward_runs_on_the_mfma_path(case):
    """Shapes meant for the matrix-core kernels must not silently fall back to the direct ones
    (conv groups and deformable groups included)."""
    assert _check(case, torch.float32, "auto")[1] == "mfma"


@pytest.mark.parametrize("name", ["mfma_split_mdcn2d_dg8_c128_o32", "mfma_split_dcn3d_g2_dg8_c128_o32"])
def test_forward_of_small_deformable_groups_runs_on_the_mfma_path(name):
    """C_in/DG = 16 inside conv groups of 128 / 64 channels: per-slice forwards on the matrix-core kernels,
    summed into the conv group's output channels (the first slice carries the bias)."""
    from tests.cases import CASE_BY_NAME
    assert _check(CASE_BY_NAME[name], torch.float32, "auto")[0] == "mfma"


def test_fp16_grouped_mfma_backward():
    from tests.cases import CASE_BY_NAME
    assert _check(CASE_BY_NAME["cfg3s_mdcn2d_c256_g32_dg4_10x10"], torch.float16, "auto")[1] == "mfma"


@pytest.mark.parametrize("case", [c for c in CASES if c["tier"] == "small"], ids=lambda c: c["name"])
def test_fp64(case):
    _check(case, torch.float64, "auto")


def test_large_kernel_volume_uses_the_big_lds_tile():
    """7x7x7 taps (K = 343): the direct path's fp64 backward weight tile is 88 KB, above the 64 KB
    default of dynamic LDS (the reference handles any kernel volume; gfx950 allows 160 KB)."""
    from tests.cases import D3, M3, _c
    _check(_c("dcn3d_k7", D3, 1, 2, 2, (6, 6, 6), 7, padding=3, seed=61), torch.float64, "auto")
    _check(_c("mdcn3d_k7", M3, 1, 2, 4, (5, 6, 5), 7, padding=3, seed=62), torch.float32, "auto")


@pytest.mark.parametrize("case", [c for c in CASES if c["tier"] == "small"][::2] +
                         [c for c in CASES if c["tier"] == "medium"][:2], ids=lambda c: c["name"])
def test_fp16(case):
    _check(case, torch.float16, "auto")


@pytest.mark.parametrize("name", ["dcn2d_s2_g2_dg2", "mfma_split_dcn2d_g2_dg4_c128_o64"])
def test_accumulate_semantics(name):
    """Backward entry points accumulate into caller buffers (deformable_conv.cu:327-333) -- also when the
    call runs as per-deformable-group slices through workspace copies."""
    from tests.cases import CASE_BY_NAME
    from tests.util import tup
    case = CASE_BY_NAME[name]
    t = make_inputs(case, dtype=torch.float32, device="cuda")
    _, g1, _ = run_product(case, t, "auto")
    from modulated_deform_conv_amd import MDCONV_CUDA as M
    gi, gw = torch.ones_like(t["input"]), torch.ones_like(t["weight"])
    gb, goff = torch.ones_like(t["bias"]), torch.ones_like(t["offset"])
    k, s, p, d = (tup(case[x], 2) for x in ("k", "stride", "padding", "dilation"))
    M.deform_conv2d_backward_cuda(t["input"], t["weight"], t["bias"], t["offset"], gi, gw, gb, goff,
                                  t["grad_output"], *(k + s + p + d), case["groups"], case["dgroups"],
                                  case["in_step"], True)
    for got, base in ((gi, g1["grad_input"]), (gw, g1["grad_weight"]), (gb, g1["grad_bias"]),
                      (goff, g1["grad_offset"])):
        assert_close("accumulate", got - 1, base, 1e-4)


def test_error_behaviour_on_device():
    from modulated_deform_conv_amd import MDCONV_CUDA as M
    x = torch.randn(2, 4, 6, 6, device="cuda")
    w = torch.randn(4, 4, 3, 3, device="cuda")
    off = torch.zeros(2, 18, 6, 6, device="cuda")
    m = torch.ones(2, 9, 6, 6, device="cuda")
    b = x.new_empty(0)
    geo = (3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 64, False)
    with pytest.raises(RuntimeError, match="contiguous"):
        M.modulated_deform_conv2d_forward_cuda(x.transpose(2, 3), w, b, off, m, *geo)
    with pytest.raises(RuntimeError, match="kernel shape"):
        M.modulated_deform_conv2d_forward_cuda(x, w, b, off, m, 5, 5, *geo[2:])
    with pytest.raises(RuntimeError, match="channels"):
        M.modulated_deform_conv2d_forward_cuda(x, w, b, off, m, *geo[:8], 2, 1, 64, False)
    with pytest.raises(RuntimeError, match="offset shape"):
        M.modulated_deform_conv2d_forward_cuda(x, w, b, off[:, :10].contiguous(), m, *geo)
    with pytest.raises(RuntimeError, match="in_step"):
        M.modulated_deform_conv2d_forward_cuda(x, w, b, off, m, *geo[:10], 0, False)
    out = M.modulated_deform_conv2d_forward_cuda(x, w, b, off, m, *geo)
    assert out.shape == (2, 4, 6, 6)


def test_batch_chunking_matches_unchunked():
    """Calls whose tensors exceed the 32-bit buffer range are cut into batch chunks inside the C ABI
    (the successor of the reference's in_step loop).  Force tiny chunks in a subprocess and compare
    with the oracle."""
    import os
    import subprocess
    import sys
    code = r"""
import torch, sys
sys.path.insert(0, %r)
from tests.cases import CASE_BY_NAME, make_inputs
from tests.util import run_product, run_oracle, assert_close
from modulated_deform_conv_amd import _capi
for name, dt, tol in (("cfg2s_mdcn2d_c64_28x28_b4", torch.float32, 1e-4), ("cfg2s_mdcn2d_c64_28x28_b4", torch.float16, 5e-3),
                      ("cfg4s_dcn3d_c16_12cubed_b2", torch.float32, 1e-4)):
    case = CASE_BY_NAME[name]
    t = make_inputs(case, dtype=dt, device="cuda")
    out, g, paths = run_product(case, t, "mfma")
    assert paths == ["mfma", "mfma"], paths
    wo, w = run_oracle(case, t, torch.float32)
    assert_close("output", out, wo, tol)
    for k in g:
        if w[k] is not None:
            assert_close(k, g[k], w[k], tol)
print("CHUNK_OK")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MDCONV_CHUNK_LIMIT_BYTES=str(4 * 1024 * 1024))   # 2 or 1 images per chunk at these sizes
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert "CHUNK_OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("name, dtype, path", [
    ("dcn2d_s2_g2_dg2", torch.float32, "direct"),
    ("mdcn3d_dil2_dg2", torch.float16, "direct"),
    ("cfg2s_dcn2d_c64_28x28_b4", torch.float32, "auto"),
    ("cfg4s_dcn3d_c16_12cubed_b2", torch.float32, "auto"),
    ("mfma_mdcn3d_g2_dg2_c128_o32", torch.float32, "auto"),
    ("mfma_mdcn3d_g2_dg2_c128_o32", torch.float16, "auto"),
    ("mfma_dcn2d_c512_o32_7x5", torch.float32, "auto"),
    ("mfma_split_dcn2d_g2_dg4_c128_o64", torch.float32, "auto"),
    ("mfma_split_mdcn3d_dg2_c32_o32", torch.float32, "auto"),
    ("mfma_split_mdcn3d_dg2_c32_o32", torch.bfloat16, "auto"),
    ("mfma_pad_dcn3d_dg2_c160_o32", torch.float32, "auto"),     # one padded problem: gradients copied back from padded buffers
    ("mfma_pad_dcn2d_dg2_c16_o16", torch.float32, "auto"),
    ("mfma_pad_dcn3d_dg2_c160_o32", torch.float16, "auto"),     # group-padded layout on the native 16-bit kernels (80 -> 128)
    ("mfma_pad_mdcn3d_dg3_c72_o40_dil2", torch.float16, "auto"),
    ("mfma_padc_dcn3d_c48_o24_s2", torch.float32, "auto"),       # channel-padded plan (one deformable group)
    ("mfma_padt_dcn3d_c64_o8_576px", torch.float32, "auto"),     # output channels padded to 16: grad_bias through a padded buffer
    ("mfma_padt_dcn2d_c8_o5_9408px", torch.float32, "auto"),
    ("mfma_padt_dcn3d_c3_o5_k2", torch.float16, "auto"),
    ("mfma_padn_dcn3d_c20_o24", torch.float32, "auto"),          # backward only padded (C_in not a multiple of 8)
    ("mfma_padn_dcn3d_c24_dg2_o8", torch.float32, "auto"),       # deformable groups AND output channels padded
    ("mfma_padg_dcn3d_g4_c32_o8", torch.float32, "auto"),        # conv groups: input and output channels padded per group
    ("mfma_padg_dcn2d_g3_c36_o36_s2", torch.float32, "auto"),
    ("mfma_padg_dcn2d_g3_c36_o36_s2", torch.float16, "auto"),
    ("mfma_padgd_dcn2d_g2_dg4_c48_o32", torch.float32, "auto"),   # conv groups and deformable groups padded together
    ("mfma_padgd_dcn3d_g2_dg2_c40_o8", torch.float32, "auto"),
])
def test_overwrite_mode_writes_every_gradient_element(name, dtype, path):
    """mdconv_set_accumulate(0): the caller-allocated backward entry points must WRITE every element
    of every gradient (buffers pre-filled with NaN) and give the accumulate-into-zeros result."""
    from modulated_deform_conv_amd import MDCONV_CUDA as M, _capi
    from tests.cases import CASE_BY_NAME, ndim
    from tests.util import tup
    case = CASE_BY_NAME[name]
    t = make_inputs(case, dtype=dtype, device="cuda")
    _, want, _ = run_product(case, t, path)          # accumulate into zeros (reference semantics)
    nd = ndim(case)
    k, s, p, d = (tup(case[x], nd) for x in ("k", "stride", "padding", "dilation"))
    geo = k + s + p + d + (case["groups"], case["dgroups"], case["in_step"], case["bias"])
    x, w, off, m, go = t["input"], t["weight"], t["offset"], t["mask"], t["grad_output"]
    b = t["bias"] if case["bias"] else x.new_empty(0)
    nan = lambda ref: torch.full_like(ref, float("nan"))
    gi, gw, gb, goff = nan(x), nan(w), nan(b), nan(off)
    gm = nan(m) if m is not None else None
    fn = getattr(M, "%sdeform_conv%dd_backward_cuda" % ("modulated_" if m is not None else "", nd))
    prev = _capi.set_path(path)
    try:
        with _capi.overwrite_grads():
            if m is None:
                fn(x, w, b, off, gi, gw, gb, goff, go, *geo)
            else:
                fn(x, w, b, off, m, gi, gw, gb, goff, gm, go, *geo)
    finally:
        _capi.set_path(prev)
    got = {"grad_input": gi, "grad_weight": gw, "grad_offset": goff, "grad_mask": gm,
           "grad_bias": gb if case["bias"] else None}
    # fp16: the two runs round differently ordered sums to 11 bits (atomics / CSR list order)
    tol, elem_tol = (1e-5, None) if dtype == torch.float32 else (2e-3, 1e-2)
    if dtype == torch.bfloat16:
        tol, elem_tol = 1.6e-2, 8e-2
    for key, g in got.items():
        if g is None or want[key] is None:
            continue
        assert torch.isfinite(g).all(), key
        assert_close(key, g, want[key], tol, elem_tol)


@pytest.mark.parametrize("path", ["mfma", "direct"])
def test_fp32_non_finite_border_pixel_is_not_read(path):
    """1x1 kernel, x-offset -0.5 everywhere: output column 0 samples between columns -1 (outside, never
    read by the reference) and 0; the fp32 kernels fetch the pair (column 0, column 1) and give column 1
    the weight 0.  With Inf in input column 1 the reference's column 0 stays finite.  Round 4: the forward's
    pair loads keep their 8 bytes, but the waves that hold such a sample select the unread element away before the
    multiply (mfma_fwd.hip, `bad` lane masks); pairs with nothing to read, every channels-last gather and the
    shape-generic kernels park / skip the corner like the 16-bit kernels do (tests/test_gpu_hp.py)."""
    case = CASE_BY_NAME["mfma_mdcn2d_k1_c64_o32"]
    t = make_inputs(case, dtype=torch.float32, device="cuda")
    t["offset"].zero_()
    t["offset"][:, 0] = 0.25
    t["offset"][:, 1] = -0.5
    t["input"][0, :, :, 1] = float("inf")
    out, _, _ = run_product(case, t, path)
    want_out, _ = run_oracle(case, t, torch.float32)
    assert torch.isfinite(want_out[0, :, :, 0]).all() and not torch.isfinite(want_out[0, :, :, 1]).any()
    assert torch.equal(torch.isfinite(out.cpu()), torch.isfinite(want_out))


def _inf_case(name):
    from tests.cases import D2, D3, M2, _c
    return {
        # channels-last backward (3-D always; 2-D from 8 k output pixels): every gather addresses corners one by one
        "dcn3d_c64_cl": _c("inf_dcn3d_c64", D3, 1, 64, 32, (5, 6, 5), 3, seed=161),
        "mdcn2d_c64_cl": _c("inf_mdcn2d_c64", M2, 6, 64, 32, (56, 56), 3, seed=162),
        # small 2-D shape: the NCHW backward kernels (8-byte pair loads in GEMM-1's drain and in GEMM-2)
        "mdcn2d_c64_small": _c("inf_mdcn2d_c64_small", M2, 2, 64, 32, (9, 11), 3, seed=163),
        "mdcn2d_c48_small": _c("inf_mdcn2d_c48_small", M2, 2, 48, 40, (9, 11), 3, seed=164),
        "dcn3d_c32_nchw": _c("inf_dcn3d_c32", D3, 1, 32, 32, (5, 6, 5), 3, seed=165),
    }[name]


@pytest.mark.parametrize("name,path", [
    ("dcn3d_c64_cl", "mfma"), ("mdcn2d_c64_cl", "mfma"), ("mdcn2d_c64_small", "direct"),
    # the NCHW backward kernels (8-byte pair loads in GEMM-1's drain and in GEMM-2; a strict xfail until round 5):
    # the element that comes along with a wanted neighbour is selected away, 2-D and (C_in not a multiple of 64) 3-D
    ("mdcn2d_c64_small", "mfma"), ("mdcn2d_c48_small", "mfma"), ("dcn3d_c32_nchw", "mfma")])
def test_fp32_backward_with_a_non_finite_border_pixel(name, path):
    """Inf in ONE channel of one border pixel: the gradients that the reference keeps finite stay finite -- a corner
    outside the image is never read (mdeformable_conv.cu:256-267), and neither is the in-image neighbour that a
    paired or clamped load would fetch next to it."""
    case = _inf_case(name)
    t = make_inputs(case, dtype=torch.float32, device="cuda")
    nd = len(case["in_sz"])
    t["input"][(0, 3) + (0,) * (nd - 1) + (1,)] = float("inf")     # second pixel of the first row: a pair's "other" element
    t["input"][(0, 5) + (0,) * nd] = float("inf")                   # the corner pixel itself
    out, grads, paths = run_product(case, t, path)
    want_out, want = run_oracle(case, t, torch.float32)
    assert paths[1] == path
    assert torch.equal(torch.isfinite(out.cpu()), torch.isfinite(want_out))
    for k in ("grad_offset", "grad_mask", "grad_input", "grad_weight"):
        if want[k] is not None:
            assert torch.equal(torch.isfinite(grads[k].cpu()), torch.isfinite(want[k])), k


@pytest.mark.parametrize("O", [16, 40])
def test_forward_tail_tap_ranges_on_a_ragged_last_tile(O):
    """More tiles than resident workgroup slots, a last dispatch round that is mostly empty and a ragged last tile:
    the forward cuts the leftover tiles into tap ranges and adds the partial tiles up (mfma_fwd.hip, fwd_tail_plan).
    C_out = 16 / 40 select the 64 x 128 tile (800 tiles of 128 pixels on 768 slots); 102 396 pixels are not a
    multiple of 128, so the last tail tile is partly outside the batch; C_out = 40 also leaves padded rows."""
    from tests.cases import M2, _c
    case = _c("tail_mdcn2d_c16", M2, 4, 16, O, (161, 159), 3, seed=171)
    t = make_inputs(case, dtype=torch.float32, device="cuda")
    out, grads, paths = run_product(case, t, "mfma")
    assert paths == ["mfma", "mfma"]
    want_out, want = run_oracle(case, t, torch.float32)
    assert_close("output", out, want_out, 1e-4)
    for k, g in grads.items():
        if want[k] is not None:
            assert_close(k, g, want[k], 1e-4)
