"""Oracle-independent analytic pins at non-zero offsets (tests/analytic_pins.py): integer-shift identity and
linear-ramp closed forms for all four ops -- asserted on the CPU oracle (so the checker itself is pinned at
non-zero offsets by something that shares no code and no reading of SURVEY.md with it) and on the HIP kernels
(both kernel paths, every dtype)."""
import pytest
import torch

import oracle
from tests.analytic_pins import (grad_input_moments, integer_shift, linear_ramp, ramp_geometries,
                                 shift_geometries)
from tests.util import assert_close

SHIFT = shift_geometries()
RAMP = ramp_geometries()


def _oracle_run(geo, t):
    args = (geo.stride, geo.padding, geo.dilation, geo.groups, geo.dgroups, 64)
    out = oracle.forward(geo.op, t["input"], t["weight"], t["bias"], t["offset"], t["mask"], *args,
                         dtype=torch.float64)
    g = oracle.backward(geo.op, t["input"], t["weight"], t["bias"], t["offset"], t["mask"], t["grad_output"],
                        *args, dtype=torch.float64)
    return out, g


def _compare(out, grads, want, tol):
    assert_close("output", out.double(), want["output"], tol)
    for k, w in want.items():
        if k != "output" and w is not None:
            assert_close(k, grads[k].double(), w, tol)


def _moment_close(got, want, tol):
    # first moments carry a factor of up to (size - 1): compare relative to the tensor's scale
    scale = max(1.0, want.abs().max().item())
    assert ((got - want).abs().max() / scale).item() <= tol, ((got - want).abs().max().item(), scale)


# ------------------------------------------------------------------ the oracle itself (CPU)
@pytest.mark.parametrize("name,geo", SHIFT, ids=[n for n, _ in SHIFT])
def test_oracle_integer_shift_identity(oracle_lib, name, geo):
    t, want = integer_shift(geo, seed=500)
    assert t["offset"].abs().max() >= 1 and (t["offset"] == t["offset"].round()).all()
    out, g = _oracle_run(geo, t)
    _compare(out, g, want, 1e-11)


@pytest.mark.parametrize("name,geo", RAMP, ids=[n for n, _ in RAMP])
def test_oracle_linear_ramp_closed_forms(oracle_lib, name, geo):
    t, want, moments = linear_ramp(geo, seed=600)
    out, g = _oracle_run(geo, t)
    _compare(out, g, want, 1e-11)
    _moment_close(grad_input_moments(g["grad_input"]), moments, 1e-11)


def test_ramp_pins_the_axis_order(oracle_lib):
    """The pin is sensitive: swapping the h / w offset channels of one tap changes the expected grad_offset by far
    more than the tolerance (a guard against a scenario that is accidentally symmetric)."""
    name, geo = RAMP[0]
    t, want, _ = linear_ramp(geo, seed=600)
    swapped = want["grad_offset"].clone()
    swapped[:, [0, 1]] = swapped[:, [1, 0]]
    assert (swapped - want["grad_offset"]).abs().max() > 1e-2


# ------------------------------------------------------------------ the HIP kernels (GPU)
GPU_TOL = {torch.float32: 1e-4, torch.float64: 1e-10, torch.float16: 5e-3, torch.bfloat16: 3e-2}


def _gpu_cases(geos):
    """(name, geo, dtype, path): fp32 on both paths everywhere; fp64 on the small shapes; fp16 / bf16 everywhere
    (the native 16-bit kernels take the matrix-sized shapes, the rest runs through their fallbacks)."""
    out = []
    for name, geo in geos:
        out.append((name, geo, torch.float32, "direct"))
        out.append((name, geo, torch.float32, "auto"))
        if geo.C <= 8:
            out.append((name, geo, torch.float64, "auto"))
        out.append((name, geo, torch.float16, "auto"))
        out.append((name, geo, torch.bfloat16, "auto"))
    return out


def _gpu_id(c):
    return "%s-%s-%s" % (c[0], str(c[2]).replace("torch.", ""), c[3])


def _product(geo, t, dtype, path):
    from tests.util import run_product
    td = {k: (None if v is None else v.to("cuda", dtype)) for k, v in t.items()}
    out, grads, paths = run_product(geo.case("pin"), td, path)
    torch.cuda.synchronize()
    return out, grads, paths


@pytest.mark.gpu
@pytest.mark.parametrize("c", _gpu_cases(SHIFT), ids=_gpu_id)
def test_hip_integer_shift_identity(c):
    name, geo, dtype, path = c
    t, want = integer_shift(geo, seed=500, dtype=dtype)
    out, grads, _ = _product(geo, t, dtype, path)
    _compare(out, grads, want, GPU_TOL[dtype])


@pytest.mark.gpu
@pytest.mark.parametrize("c", _gpu_cases(RAMP), ids=_gpu_id)
def test_hip_linear_ramp_closed_forms(c):
    name, geo, dtype, path = c
    t, want, moments = linear_ramp(geo, seed=600, dtype=dtype)
    out, grads, _ = _product(geo, t, dtype, path)
    _compare(out, grads, want, GPU_TOL[dtype])
    _moment_close(grad_input_moments(grads["grad_input"]), moments, GPU_TOL[dtype])


@pytest.mark.gpu
def test_matrix_shapes_reach_the_matrix_kernels():
    """The pins above must exercise the MFMA kernels, not only the shape-generic ones."""
    for geos, build in ((SHIFT, lambda g: integer_shift(g, 500, torch.float32)[0]),
                        (RAMP, lambda g: linear_ramp(g, 600, torch.float32)[0])):
        for name, geo in geos:
            if geo.C >= 64:
                _, _, paths = _product(geo, build(geo), torch.float32, "auto")
                assert paths == ["mfma", "mfma"], (name, paths)
