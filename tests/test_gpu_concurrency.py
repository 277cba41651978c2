"""Host threads on their own streams call the library at the same time (autograd's backward thread, data-parallel worker
threads, user streams): ctypes releases the GIL, so the host side of several calls -- plan caches, the fork-stream and
weights-ready tables, the workspace layout -- really runs concurrently, and their kernels share the chip.  Every call must
return what the same call returns alone: output, grad_weight, grad_bias, grad_offset and grad_mask bit for bit
(INTEGRATION.md "Reproducibility"), grad_input to rounding (its addition order follows integer-atomic arrival)."""
import threading

import pytest
import torch

from tests.cases import case_f32_wide, case_hp_wide, make_inputs, D3, M2, M3, _c
from tests.util import rel_err, run_product

pytestmark = pytest.mark.gpu

ITERS = 20


def _workers():
    """(case, dtype): fp32 matrix-core kernels in 2-D and 3-D (forked backward), the native 16-bit kernels (fused and
    pixel-stationary backward), a shape of the generic kernels -- different kernel families side by side."""
    return [
        (_c("conc_mdcn2d_c64", M2, 4, 64, 64, (28, 28), 3, seed=901), torch.float32),
        (_c("conc_dcn3d_c32", D3, 2, 32, 32, (8, 10, 10), 3, seed=902), torch.float32),
        (_c("conc_mdcn2d_f16_g4_dg2", M2, 4, 128, 128, (20, 20), 3, groups=4, dgroups=2, seed=903), torch.float16),
        (_c("conc_mdcn3d_f16_c128", M3, 2, 128, 128, (6, 12, 12), 3, seed=904), torch.float16),
        (case_f32_wide(104), torch.float32),
        (case_hp_wide(108), torch.bfloat16),
        (_c("conc_generic_c6", M2, 3, 6, 5, (9, 9), 3, seed=905), torch.float32),
        (_c("conc_mdcn2d_c64_again", M2, 4, 64, 64, (28, 28), 3, seed=901), torch.float32),   # same shape as worker 0
    ]


def _same(name, got, want, exact):
    if got is None or want is None:
        assert got is None and want is None, name
        return
    if exact:
        assert torch.equal(got, want), "%s differs from the serial call (scaled %.2e)" % (name, rel_err(got.float(), want.float()))
    else:
        tol = 1e-5 if got.dtype == torch.float32 else 4e-3
        assert rel_err(got.float(), want.float()) <= tol, "%s: scaled %.2e" % (name, rel_err(got.float(), want.float()))


def test_concurrent_host_threads_on_their_own_streams():
    workers = _workers()
    inputs = [make_inputs(c, dtype=dt, device="cuda") for c, dt in workers]
    serial = [run_product(c, t) for (c, _), t in zip(workers, inputs)]
    torch.cuda.synchronize()
    errors = []
    start = threading.Barrier(len(workers))

    def work(i):
        try:
            case, _ = workers[i]
            out0, g0, p0 = serial[i]
            stream = torch.cuda.Stream()
            start.wait()
            with torch.cuda.stream(stream):
                for it in range(ITERS):
                    out, g, p = run_product(case, inputs[i])
                    stream.synchronize()
                    _same("%s output (iteration %d)" % (case["name"], it), out, out0, True)
                    for k in ("grad_weight", "grad_bias", "grad_offset", "grad_mask"):
                        _same("%s %s (iteration %d)" % (case["name"], k, it), g[k], g0[k], True)
                    _same("%s grad_input (iteration %d)" % (case["name"], it), g["grad_input"], g0["grad_input"], False)
        except BaseException as e:   # noqa: BLE001 -- reported by the main thread
            errors.append("%s: %s: %s" % (workers[i][0]["name"], type(e).__name__, str(e).split("\n")[0][:300]))
            try:
                start.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(workers))]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not any(th.is_alive() for th in threads), "a worker thread hangs"
    torch.cuda.synchronize()
    assert not errors, errors
