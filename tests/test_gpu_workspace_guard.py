"""No entry point may touch memory outside the workspace it asked for.

Every call of the shared case list (both fp32 kernel paths, the 16-bit kernels) and of a block of random shapes runs with
its workspace placed in the MIDDLE of a larger allocation whose two margins carry a byte pattern; after the call the
margins must be untouched.  (Round 5: `tools/fuzz_more.py` ended in a GPU memory fault on DeformConv2d C=64 DG=4 with
bias -- the split backward sized its per-slice workspace without the grad_bias stage buffer that the first slice of a
conv group appends, and wrote 32 * C_out * 4 bytes past the end.  The parity tests could not see it: the bytes landed in
whatever the caching allocator had next to the workspace.)"""
import ctypes

import pytest
import torch

from tests.cases import CASES, D2, D3, M2, M3, _c, make_inputs
from tests.util import run_product

pytestmark = pytest.mark.gpu

PAD = 1 << 20
PATTERN = 0xA5


@pytest.fixture
def guarded(monkeypatch):
    from modulated_deform_conv_amd import MDCONV_CUDA as M
    from modulated_deform_conv_amd import _capi
    touched = []

    def run_guarded(fn_name, d, backward, args_before_ws, input):
        L = _capi.lib()
        d.input_layout = int(not input.is_contiguous() and M._is_channels_last(input))
        with torch.cuda.device(input.device):
            ws_bytes = L.mdconv_workspace_bytes(ctypes.byref(d), int(backward))
            big = torch.full((ws_bytes + 2 * PAD,), PATTERN, dtype=torch.uint8, device=input.device)
            stream = torch.cuda.current_stream().cuda_stream
            rc = getattr(L, fn_name)(ctypes.byref(d), *args_before_ws, ctypes.c_void_p(big.data_ptr() + PAD),
                                     ctypes.c_size_t(ws_bytes), ctypes.c_void_p(stream))
            torch.cuda.synchronize()
            for side, region in (("below", big[:PAD]), ("above", big[PAD + ws_bytes:])):
                bad = (region != PATTERN).nonzero()
                if bad.numel():
                    touched.append("%s: %d bytes %s the workspace (%d bytes), first at %+d" % (
                        fn_name, bad.numel(), side, ws_bytes, int(bad[0]) - (PAD if side == "below" else 0)))
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (fn_name, rc, _capi.last_error()))

    monkeypatch.setattr(M, "_run", run_guarded)
    return touched


EXTRA = [
    # the shape of the round-5 fault and relatives: split backward / forward (C_in / DG = 16, 24, 32) WITH bias
    _c("guard_dcn2d_c64_dg4_o17_bias", D2, 5, 64, 17, (19, 15), 3, padding=0, dgroups=4, in_step=1, tier="medium", seed=7173,
       offset_scale=0.5),
    _c("guard_mdcn2d_c96_dg4_g1_bias", M2, 2, 96, 40, (11, 12), 3, dgroups=4, tier="medium", seed=7174),
    _c("guard_mdcn3d_c64_dg2_bias", M3, 2, 64, 24, (5, 6, 5), 3, dgroups=2, tier="medium", seed=7175),
    _c("guard_dcn3d_c128_g2_dg8_bias", D3, 1, 128, 32, (5, 5, 6), 3, groups=2, dgroups=8, tier="medium", seed=7176),
]


@pytest.mark.parametrize("case", CASES + EXTRA, ids=lambda c: c["name"])
def test_fp32_calls_stay_inside_their_workspace(case, guarded):
    t = make_inputs(case, device="cuda")
    for path in ("auto", "direct"):
        run_product(case, t, path)
    assert not guarded, guarded


def _hp_ok(case):
    return case["C"] >= 8 and case["tier"] != "small"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("case", [c for c in CASES + EXTRA if _hp_ok(c)], ids=lambda c: c["name"])
def test_16bit_calls_stay_inside_their_workspace(case, dtype, guarded):
    t = make_inputs(case, dtype=dtype, device="cuda")
    run_product(case, t, "auto")
    assert not guarded, guarded


def test_random_shapes_stay_inside_their_workspace(guarded):
    from tests.test_gpu_fuzz import _random_case, _random_hp_case
    for seed in range(40):
        case = _random_case(seed)
        run_product(case, make_inputs(case, device="cuda"), "auto")
    for seed in range(24):
        case = _random_hp_case(seed)
        run_product(case, make_inputs(case, dtype=torch.float16, device="cuda"), "auto")
    assert not guarded, guarded
