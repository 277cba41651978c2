"""Golden-vector tests: committed inputs/expected outputs under tests/golden/ (made by
tests/golden/make_golden.py from the fp64 oracle).  CPU: the oracle must still reproduce them.
GPU: the HIP path, both kernel paths, must reproduce them without the oracle in the loop."""
import glob
import os

import pytest
import torch

import oracle
from tests.util import assert_close, run_product

FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.pt")))


def _load(path):
    return torch.load(path, weights_only=False)


def test_golden_files_present():
    assert len(FILES) >= 9


@pytest.mark.parametrize("path", FILES, ids=lambda p: os.path.basename(p)[:-3])
def test_oracle_reproduces_golden(oracle_lib, path):
    blob = _load(path)
    case, t, want = blob["case"], blob["inputs"], blob["expected"]
    args = (case["stride"], case["padding"], case["dilation"], case["groups"], case["dgroups"], case["in_step"])
    x = {k: t.get(k).double() if t.get(k) is not None else None for k in ("input", "weight", "bias", "offset", "mask", "grad_output")}
    out = oracle.forward(case["op"], x["input"], x["weight"], x["bias"], x["offset"], x["mask"], *args)
    g = oracle.backward(case["op"], x["input"], x["weight"], x["bias"], x["offset"], x["mask"], x["grad_output"], *args)
    tol = 1e-12 if want["output"].dtype == torch.float64 else 1e-6
    assert_close("output", out, want["output"].double(), tol)
    for k, v in g.items():
        if v is not None:
            assert_close(k, v, want[k].double(), tol)


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=lambda p: os.path.basename(p)[:-3])
@pytest.mark.parametrize("kpath", ["direct", "auto"])
def test_hip_reproduces_golden(path, kpath):
    blob = _load(path)
    case, want = blob["case"], blob["expected"]
    t = {k: blob["inputs"].get(k) for k in ("input", "weight", "bias", "offset", "mask", "grad_output")}
    t = {k: (None if v is None else v.cuda()) for k, v in t.items()}
    out, grads, _ = run_product(case, t, kpath)
    assert_close("output", out, want["output"], 1e-4)
    for k, v in grads.items():
        if v is not None:
            assert_close(k, v, want[k], 1e-4)
