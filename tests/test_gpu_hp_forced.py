"""The lane = pixel predecessors of the native 16-bit kernels (hp_fwd.hip, hp_bwd.hip) are still live
product code: hp_host.hip selects hp_bwd when a shape has more than 4 deformable groups or needs
more than 160 KB of LDS in hp_bwd2, and hp_fwd when a 64-channel K stage would straddle
deformable groups.  Only a few shapes reach them on their own, so the whole 16-bit case list runs
again in a child process with MDCONV_HP_FWD=1 / MDCONV_HP_BWD=1 (read once per process) and is
compared with the oracle as usual.  A second child forces uneven batch chunks through the 16-bit
path (MDCONV_CHUNK_LIMIT_BYTES): pointer offsets per chunk, the fp32 running grad_weight and a
channels-last input -- the batch-chunk loop of hp_host.hip that only multi-GiB calls reach."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parity_with_lane_per_pixel_16bit_kernels_forced():
    env = dict(os.environ, MDCONV_HP_FWD="1", MDCONV_HP_BWD="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_hp.py", "-m", "gpu", "-q", "-x",
                        "-k", "test_hp_fp16 or test_hp_bf16 or accumulate_and_overwrite or non_finite"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_parity_with_the_pixel_stationary_backward_forced():
    """Small shapes take the tap-stationary hp_bwd2 by default since round 6 (hp_host.hip, use_bwd3: up to one 128-pixel tile
    per CU); MDCONV_HP_BWD=4 keeps hp_bwd3 wherever it is supported, so that every instance the case list reaches -- staged
    slabs, A fragments from global memory, 1 / 2 / 4 deformable groups of 16-128 channels -- is compared with the oracle."""
    env = dict(os.environ, MDCONV_HP_BWD="4")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_hp.py", "tests/test_gpu_fuzz.py", "-m", "gpu", "-q", "-x",
                        "-k", "test_hp_fp16 or test_hp_bf16 or accumulate_and_overwrite or 16bit_random_shapes"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


CHUNK_CODE = r"""
import sys
sys.path.insert(0, %r)
import torch
from tests.cases import _c, make_inputs, M2, M3
from tests.util import assert_close, run_oracle, run_product
from modulated_deform_conv_amd import _capi
cases = {"2d_fp16": (_c("chunk_mdcn2d_c64_o64", M2, 20, 64, 64, (24, 20), 3, seed=141), torch.float16, False),
         "2d_bf16_cl": (_c("chunk_mdcn2d_c64_o64_cl", M2, 20, 64, 64, (24, 20), 3, seed=141), torch.bfloat16, True),
         "3d_fp16": (_c("chunk_mdcn3d_c32_o32", M3, 30, 32, 32, (4, 8, 8), 3, seed=142), torch.float16, False),
         # group-padded layout (2 groups of 24 channels run as 2 x 32): chunk pointers step by the CALLER's 48 channels
         "2d_fp16_pad": (_c("chunk_mdcn2d_c48_dg2_o64", M2, 20, 48, 64, (24, 20), 3, dgroups=2, seed=143), torch.float16, False)}
case, dtype, cl = cases[sys.argv[1]]
t = make_inputs(case, dtype=dtype, device="cuda")
if cl:
    t["input"] = t["input"].contiguous(memory_format=torch.channels_last)
out, g, _ = run_product(case, t, "auto")
torch.cuda.synchronize()
assert _capi.last_kernels() == "hp", _capi.last_kernels()
wo, w = run_oracle(case, {k: (None if v is None else v.float().contiguous()) for k, v in t.items()}, torch.float32)
tol = 5e-3 if dtype == torch.float16 else 3e-2
assert_close("output", out.float(), wo, tol)
for k in g:
    if w[k] is not None:
        assert_close(k, g[k].float(), w[k], tol)
print("HP_CHUNK_OK")
"""


# The limit bounds a chunk's channels-last input copy AND one image's grad_col rows, so it cannot
# go below the latter: 2-D 9 * 480 * 64 * 2 = 552 960 B (input copy 61 440 B per image -> 9 images
# per chunk, B = 20 -> chunks of 9, 9, 2); 3-D 27 * 256 * 32 * 2 = 442 368 B (16 384 B per image ->
# 27 per chunk, B = 30 -> 27 + 3).  The last chunk is shorter in every case.
@pytest.mark.parametrize("which, limit", [("2d_fp16", 600_000), ("2d_bf16_cl", 600_000), ("3d_fp16", 450_000), ("2d_fp16_pad", 600_000)])
def test_16bit_batch_chunk_loop_with_uneven_chunks(which, limit):
    env = dict(os.environ, MDCONV_CHUNK_LIMIT_BYTES=str(limit))
    r = subprocess.run([sys.executable, "-c", CHUNK_CODE % ROOT, which], env=env, capture_output=True,
                       text=True, timeout=600)
    assert "HP_CHUNK_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


C2I_CODE = r"""
import sys
sys.path.insert(0, %r)
import torch
from tests.cases import _c, make_inputs, D3, M3
from tests.util import run_product
from tests.cases import M2
cases = {"3d": _c("c2i_mdcn3d_s2_c64", M3, 2, 64, 32, (7, 8, 9), 3, stride=2, seed=151),   # stride 2: sums with cancellation
         "2d": _c("c2i_mdcn2d_s2_c64_dg2", M2, 2, 64, 32, (13, 12), 3, stride=2, dgroups=2, seed=152)}
case = cases[sys.argv[2]]
t = make_inputs(case, dtype=torch.bfloat16, device="cuda")
_, g, _ = run_product(case, t, "auto")
torch.cuda.synchronize()
from modulated_deform_conv_amd import _capi
assert _capi.last_kernels() == "hp", _capi.last_kernels()
torch.save(g["grad_input"].float().cpu(), sys.argv[1])
print("C2I_OK")
"""


@pytest.mark.parametrize("which", ["3d", "2d"])
def test_bf16_two_pass_gather_rounds_once_like_the_one_pass_gather(tmp_path, which):
    """Round-3 advisor (medium): the two-pass grad_input gather rounded every per-anchor partial sum to bf16 and
    the stencil sum again through a 16-bit LDS tile, where the one-pass kernel accumulates in fp32 and rounds once.
    bf16 partial sums are fp32 now (hp_col2im.hip, SumStore) and the combine tile is fp32: both kernels round each
    grad_input element ONCE from an fp32 sum of the same terms (in a different order), so they agree to one bf16
    ulp of the element plus the fp32 reordering noise -- on a 3-D stride-2 case, whose sums cancel.  Round 5: the same
    in 2-D (bf16 tensors keep fp32 list weights there too; a matrix-core partial-sums kernel with bf16 weights failed
    exactly this test and is used for fp16 tensors only)."""
    import torch
    res = {}
    for mode in ("2", "1"):
        path = str(tmp_path / ("gi_%s.pt" % mode))
        env = dict(os.environ, MDCONV_HP_C2I=mode)
        r = subprocess.run([sys.executable, "-c", C2I_CODE % ROOT, path, which], env=env, capture_output=True, text=True,
                           timeout=600)
        assert "C2I_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
        res[mode] = torch.load(path)
    two, one = res["2"].double(), res["1"].double()
    rms = one.pow(2).mean().sqrt().item()
    # one bf16 ulp = 2^-7 of the element's binade: |diff| <= 2^-7 |value| + a sliver of the tensor's scale for the
    # elements whose fp32 sums straddle a rounding boundary near zero
    bound = 2.0 ** -7 * one.abs() + 2.0 ** -12 * rms
    assert ((two - one).abs() <= bound).all(), ((two - one).abs() - bound).max().item()
    assert (two != one).float().mean().item() < 0.2     # and most elements are bit-identical
