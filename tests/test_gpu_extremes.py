"""Shapes at the edges of the supported range (tests/cases.py EXTREME_*; campaign: tools/extremes.py,
profiles/r05_experiments.md 26): hundreds of taps, axes of 65 536 pixels, thousands of channels, thousands of images of one
output position, offsets far outside the image.  fp32: default kernel selection against the shape-generic kernels (two
independent implementations) and, where it takes seconds, the CPU oracle; fp16: the native kernels against the fp32 kernels on
the same rounded values.  The two multi-second shapes of the campaign (3000 x 3000, 70 000 images) stay in the tool."""
import pytest
import torch

from tests.cases import EXTREME_F32, EXTREME_HP, make_inputs
from tests.util import assert_close, run_oracle, run_product

pytestmark = pytest.mark.gpu

_SLOW = ("x_2d_3000x3000_c16", "x_b70000_1x1_k1", "xh_2d_1500x1500_c32")
# fp32 coordinates beyond 2^15 pixels: the reference's `(p + 1 - high)` rounds once more where p + 1 crosses a power of two
# (mdeformable_conv.cu:288).  Since round 6 the kernels restate that expression (mdconv_common.hpp, make_tap), so the 1 x 65536
# image is held to the standard tolerance like every other shape (rounds 2-5: `p - low`, 5e-3 here)
_ORACLE_TOL = {}


@pytest.mark.parametrize("case", [c for c in EXTREME_F32 if c["name"] not in _SLOW], ids=lambda c: c["name"])
def test_extreme_shape_fp32(case):
    t = make_inputs(case, device="cuda")
    out_a, g_a, _ = run_product(case, t, "auto")
    out_d, g_d, _ = run_product(case, t, "direct")
    assert_close("output", out_a, out_d, 1e-4)
    for k, v in g_a.items():
        if v is not None and g_d[k] is not None:
            assert_close(k, v, g_d[k], 1e-4)
    if case["B"] * case["C"] * case["O"] * int(torch.tensor(case["in_sz"]).prod()) < 3e7:
        tol = _ORACLE_TOL.get(case["name"], 1e-4)
        want_out, want = run_oracle(case, t, torch.float32)
        assert_close("output/oracle", out_a, want_out, tol)
        for k, v in g_a.items():
            if v is not None and want[k] is not None:
                assert_close(k + "/oracle", v, want[k], tol)


@pytest.mark.parametrize("case", [c for c in EXTREME_HP if c["name"] not in _SLOW], ids=lambda c: c["name"])
def test_extreme_shape_fp16(case):
    t = make_inputs(case, dtype=torch.float16, device="cuda")
    out, g, _ = run_product(case, t, "auto")
    out32, g32, _ = run_product(case, {k: (None if v is None else v.float()) for k, v in t.items()}, "auto")
    assert_close("output", out.float(), out32, 1e-2)
    for k, v in g.items():
        if v is not None and g32[k] is not None:
            assert_close(k, v.float(), g32[k], 1e-2)
