"""Known answers derived by hand from the reference sources at integer sample positions
(tests/known_answers.py), asserted on the oracle (CPU) and on the HIP kernels (GPU)."""
import pytest
import torch

import oracle
from tests.known_answers import expected, scenario

OPS = (oracle.DCN2D, oracle.MDCN2D, oracle.DCN3D, oracle.MDCN3D)


@pytest.mark.parametrize("op", OPS, ids=lambda o: oracle.OP_NAMES[o])
def test_oracle_known_answers(op):
    nd, t = scenario(op)
    want = expected(op)
    out = oracle.forward(op, t["input"], t["weight"], t["bias"], t["offset"], t["mask"], 1, 1, 1, 1, 1, 64)
    g = oracle.backward(op, t["input"], t["weight"], t["bias"], t["offset"], t["mask"], t["grad_output"],
                        1, 1, 1, 1, 1, 64)
    assert torch.equal(out, want["output"])
    for k, w in want.items():
        if k != "output":
            assert torch.equal(g[k], w), k
    total = {2: 169, 3: 2197}[nd]
    assert out.sum().item() == total
    if op == oracle.MDCN2D:
        assert g["grad_offset"].abs().sum().item() == 52
    else:
        assert g["grad_offset"].abs().sum().item() == nd * total


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["direct", "auto"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.float16])
@pytest.mark.parametrize("op", OPS, ids=lambda o: oracle.OP_NAMES[o])
def test_hip_known_answers(op, dtype, path):
    from tests.cases import _c
    from tests.util import run_product
    nd, t = scenario(op)
    want = expected(op)
    case = _c("known", op, 1, 1, 1, (5,) * nd, 3)
    td = {k: (None if v is None else v.to("cuda", dtype)) for k, v in t.items()}
    out, grads, _ = run_product(case, td, path)
    assert torch.equal(out.float().cpu(), want["output"])
    for k, w in want.items():
        if k != "output":
            assert torch.equal(grads[k].float().cpu(), w), k
