#!/usr/bin/env python3
"""Extended random-shape campaign (developer aid, run on the GPU box): more seeds than tests/test_gpu_fuzz.py and MID-SIZE
extents (several 32-pixel tiles, ragged tile tails, batch tails), where the small fuzz shapes of the suite end.
  fp32:   matrix-core path against the shape-generic path (two independent implementations), every 4th shape also
          against the CPU oracle;    16-bit: native kernels against the oracle on the fp16 / bf16-rounded inputs.
  --wide: the geometry the case list and the other campaigns never reach -- kernel extents up to 7 (2-D) / 5 (3-D), i.e. 16 ... 125
          taps, strides and dilations up to 3, extents down to one output position, offsets up to 8 pixels.
  --dg:   16-bit shapes of the pixel-stationary backward's round-6 domain: one conv group, 1 / 2 / 4 deformable groups of 16-128
          channels on 32-256 input channels, up to 256 output channels (run it with MDCONV_HP_BWD=4 so that small shapes take hp_bwd3 too).
16-bit results are compared with the oracle whose `columns` / `grad_columns` are stored in the tensors' type, as the reference's are
(oracle.backward(intermediates=...), mdeformable_conv.cu:396-397): the rounding the kernels share with the reference is not an error.
  --pad:  16-bit shapes of the group-padded layout (deformable groups of 8 ... 120 channels run as groups of 16 ... 128)
usage: python tools/fuzz_more.py [--seconds 420] [--first 100] [--wide | --dg | --pad]      prints one line per failure and a summary."""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tests.cases import D2, D3, M2, M3, _c, case_f32_wide, case_hp_wide, make_inputs  # noqa: E402
from tests.util import assert_close, run_oracle, run_product  # noqa: E402


def case_f32(seed):
    r = random.Random(77000 + seed)
    nd = r.choice([2, 2, 3])
    modulated = r.random() < 0.6
    op = {(2, False): D2, (2, True): M2, (3, False): D3, (3, True): M3}[(nd, modulated)]
    groups = r.choice([1, 1, 1, 2, 4])
    dg = r.choice([1, 1, 1, 2, 4])
    if dg > 1:
        C = dg * r.choice([16, 32, 64, 128])
        while C % groups:
            groups //= 2
    else:
        C = r.choice([16, 24, 32, 48, 64, 72, 96, 128, 136, 192, 256])
        while C % groups:
            groups = max(1, groups // 2)
    O = r.choice([16, 17, 32, 33, 48, 64, 80, 128, 130, 256])
    O = (O + groups - 1) // groups * groups
    k = r.choice([1, 2, 3, 3, 3]) if nd == 2 else r.choice([1, 2, 3, 3])
    stride = r.choice([1, 1, 2])
    dil = r.choice([1, 1, 2])
    pad = r.choice([0, 1, dil * (k - 1) // 2 + (1 if k > 1 else 0)])
    lo = dil * (k - 1) + 1
    hi = 40 if nd == 2 else 14
    size = tuple(r.randint(max(lo, 3), hi) for _ in range(nd))
    B = r.choice([1, 2, 3, 5, 9])
    if r.random() < 0.25:   # per-axis parameters
        k = tuple(r.choice([1, 2, 3]) for _ in range(nd))
        stride = tuple(r.choice([1, 2, 3]) for _ in range(nd))
        dil = tuple(r.choice([1, 2]) for _ in range(nd))
        pad = tuple(r.choice([0, 1, 2]) for _ in range(nd))
        size = tuple(max(sz, d_ * (k_ - 1) + 1) for sz, d_, k_ in zip(size, dil, k))
    return _c("more%d" % seed, op, B, C, O, size, k, stride=stride, padding=pad, dilation=dil, groups=groups, dgroups=dg,
              in_step=r.choice([1, 64]), bias=r.random() < 0.5, tier="medium", seed=7000 + seed,
              offset_scale=r.choice([0.5, 1.0, 3.0]))


def case_hp(seed):
    r = random.Random(88000 + seed)
    nd = r.choice([2, 2, 3])
    modulated = r.random() < 0.6
    op = {(2, False): D2, (2, True): M2, (3, False): D3, (3, True): M3}[(nd, modulated)]
    dg = r.choice([1, 1, 1, 2, 4, 8])
    if dg > 1:
        C = min(256, dg * r.choice([32, 64]))
        dg = C // r.choice([c for c in (32, 64) if C % c == 0])
        groups = r.choice([1, 2])
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
    hi = 28 if nd == 2 else 10
    size = tuple(r.randint(max(lo, 3), hi) for _ in range(nd))
    size = size[:-1] + (max(size[-1], 2),)
    return _c("morehp%d" % seed, op, r.choice([1, 2, 3]), C, O, size, k, stride=stride, padding=pad, dilation=dil,
              groups=groups, dgroups=dg, in_step=64, bias=r.random() < 0.5, tier="medium", seed=8000 + seed,
              offset_scale=r.choice([0.5, 1.0, 3.0]))


def case_hp_dg(seed):
    r = random.Random(55000 + seed)
    nd = r.choice([2, 2, 3])
    modulated = r.random() < 0.6
    op = {(2, False): D2, (2, True): M2, (3, False): D3, (3, True): M3}[(nd, modulated)]
    C = r.choice([32, 64, 128, 256])
    dg = r.choice([d for d in (1, 2, 4) if C // d >= 16])
    O = r.choice([8, 24, 32, 48, 64, 100, 128, 200, 256])
    k = r.choice([1, 2, 3, 3]) if nd == 2 else r.choice([1, 2, 3])
    stride = r.choice([1, 1, 2])
    dil = r.choice([1, 1, 2])
    pad = r.choice([0, 1, dil * (k - 1) // 2 + (1 if k > 1 else 0)])
    lo = dil * (k - 1) + 1
    hi = 28 if nd == 2 else 10
    size = tuple(r.randint(max(lo, 3), hi) for _ in range(nd))
    size = size[:-1] + (max(size[-1], 2),)
    return _c("dghp%d" % seed, op, r.choice([1, 2, 3, 5]), C, O, size, k, stride=stride, padding=pad, dilation=dil,
              groups=1, dgroups=dg, in_step=64, bias=r.random() < 0.5, tier="medium", seed=5500 + seed,
              offset_scale=r.choice([0.5, 1.0, 3.0]))


def case_hp_pad(seed):
    """16-bit shapes of the group-padded layout: 2 / 3 / 4 deformable groups of 8 ... 120 channels that are not a size the kernels tile."""
    r = random.Random(66000 + seed)
    case = case_hp_dg(seed)
    dg = r.choice([2, 2, 3, 4, 4])
    cdg = r.choice([8, 12, 20, 24, 24, 40, 48, 48, 56, 72, 80, 96, 120])
    while dg * (1 << (cdg - 1).bit_length() if dg != 3 else (cdg + 31) // 32 * 32) > 256:
        cdg = r.choice([8, 12, 20, 24, 40, 48, 56])
    case.update(name="padhp%d" % seed, C=dg * cdg, dgroups=dg)
    return case


GUARD = 1 << 16
_guards = []


def guarded_zeros(like):
    """A zero tensor of `like`'s shape / dtype in the middle of a pattern-filled allocation (checked by guards_clean)."""
    n = like.numel() * like.element_size()
    big = torch.full((n + 2 * GUARD,), 0xA5, dtype=torch.uint8, device=like.device)
    view = big[GUARD:GUARD + n].view(like.dtype).view(like.shape)
    view.zero_()
    _guards.append((big, n))
    return view


def guards_clean():
    torch.cuda.synchronize()
    bad = []
    for big, n in _guards:
        for side, reg in (("below", big[:GUARD]), ("above", big[GUARD + n:])):
            if bool((reg != 0xA5).any()):
                bad.append("%d-byte tensor: written %s" % (n, side))
    _guards.clear()
    return bad


def nan_margined(t):
    """Copies of the input tensors, each in the middle of a NaN-filled allocation: a read past either end of a tensor that
    enters the arithmetic (0 * NaN included) shows up as NaN in the results."""
    out = {}
    for name, v in t.items():
        if v is None:
            out[name] = None
            continue
        pad = GUARD // v.element_size()
        big = torch.full((v.numel() + 2 * pad,), float("nan"), dtype=v.dtype, device=v.device)
        view = big[pad:pad + v.numel()].view(v.shape)
        view.copy_(v)
        out[name] = view
    return out


def all_finite(out, grads):
    bad = [] if bool(torch.isfinite(out.float()).all()) else ["output"]
    bad += [k for k, v in grads.items() if v is not None and not bool(torch.isfinite(v.float()).all())]
    return bad


def run_guarded_outputs(case, t):
    """The caller-allocated outputs of the DCN2d / DCN3d / MDCN3d entry points with guard margins (the MDCN2d entry
    points allocate their own results).  Returns the list of violated margins."""
    from tests.cases import ndim
    from tests.util import tup
    from modulated_deform_conv_amd import MDCONV_CUDA as M
    op, nd = case["op"], ndim(case)
    if op == M2:
        return []
    k, s_, p, d = (tup(case[x], nd) for x in ("k", "stride", "padding", "dilation"))
    geo = k + s_ + p + d + (case["groups"], case["dgroups"], case["in_step"], case["bias"])
    x, w, off, m, go = t["input"], t["weight"], t["offset"], t["mask"], t["grad_output"]
    b = t["bias"] if case["bias"] else x.new_empty(0)
    out = guarded_zeros(go)
    gi, gw, goff = guarded_zeros(x), guarded_zeros(w), guarded_zeros(off)
    gb = guarded_zeros(b) if case["bias"] else torch.zeros_like(b)
    if op == D2:
        M.deform_conv2d_forward_cuda(x, w, b, off, out, *geo)
        M.deform_conv2d_backward_cuda(x, w, b, off, gi, gw, gb, goff, go, *geo)
    elif op == D3:
        M.deform_conv3d_forward_cuda(x, w, b, off, out, *geo)
        M.deform_conv3d_backward_cuda(x, w, b, off, gi, gw, gb, goff, go, *geo)
    else:
        gm = guarded_zeros(m)
        M.modulated_deform_conv3d_forward_cuda(x, w, b, off, m, out, *geo)
        M.modulated_deform_conv3d_backward_cuda(x, w, b, off, m, gi, gw, gb, goff, gm, go, *geo)
    return guards_clean()


def check(name, fn):
    try:
        fn()
        return True
    except AssertionError as e:
        print("FAIL %s: %s" % (name, str(e).split("\n")[0][:300]), flush=True)
        return False


def main():
    seconds, first = 420.0, 100
    a = sys.argv[1:]
    if "--seconds" in a:
        seconds = float(a[a.index("--seconds") + 1])
    if "--first" in a:
        first = int(a[a.index("--first") + 1])
    verbose = "--verbose" in a
    gen32, gen16 = (case_f32_wide, case_hp_wide) if "--wide" in a else ((case_f32, case_hp_dg) if "--dg" in a else (case_f32, case_hp))
    if "--pad" in a:
        gen16 = case_hp_pad
    t0 = time.time()
    n = [0, 0, 0]
    bad = 0
    paths = {}
    seed = first
    while time.time() - t0 < seconds:
        # ---- fp32: matrix path vs generic path (vs oracle every 4th) ----
        case = gen32(seed)
        if verbose:
            print("run", case, flush=True)
        t = nan_margined(make_inputs(case, device="cuda"))
        out_a, g_a, p = run_product(case, t, "auto")
        if verbose:
            torch.cuda.synchronize(); print("  auto done", p, flush=True)
        paths[tuple(p)] = paths.get(tuple(p), 0) + 1
        out_d, g_d, _ = run_product(case, t, "direct")
        if verbose:
            torch.cuda.synchronize(); print("  direct done", flush=True)

        def cmp32():
            assert not all_finite(out_a, g_a), "matrix path: NaN from a read outside a tensor: %s" % all_finite(out_a, g_a)
            assert not all_finite(out_d, g_d), "generic path: NaN from a read outside a tensor: %s" % all_finite(out_d, g_d)
            assert_close("output", out_a, out_d, 1e-4)
            for k, v in g_a.items():
                if v is not None and g_d[k] is not None:
                    assert_close(k, v, g_d[k], 1e-4)
        ok = check("%s %s" % (case["name"], {k: case[k] for k in ("op", "B", "C", "O", "in_sz", "k", "stride", "padding", "dilation", "groups", "dgroups")}), cmp32)
        n[0] += 1
        viol = run_guarded_outputs(case, t)
        if viol:
            print("FAIL %s guard margins of caller tensors: %s" % (case["name"], viol), flush=True)
            ok = False
        if seed % 4 == 0 and case["B"] * case["C"] * case["O"] < 200000:
            want_out, want = run_oracle(case, t, torch.float32)

            def cmpo():
                assert_close("output/oracle", out_a, want_out, 1e-4)
                for k, v in g_a.items():
                    if v is not None and want[k] is not None:
                        assert_close(k + "/oracle", v, want[k], 1e-4)
            ok = check(case["name"] + " oracle", cmpo) and ok
            n[1] += 1
        bad += 0 if ok else 1
        # ---- 16-bit vs oracle ----
        dtype = torch.bfloat16 if seed % 3 == 0 else torch.float16
        case = gen16(seed)
        if verbose:
            print("run", dtype, case, flush=True)
        t = nan_margined(make_inputs(case, dtype=dtype, device="cuda"))
        out, grads, p = run_product(case, t, "auto")
        from modulated_deform_conv_amd import _capi
        key16 = ("16", ) + tuple(p) + ("bwd:" + _capi.last_kernels(),)   # hp = native 16-bit backward, f32 = through fp32 copies
        paths[key16] = paths.get(key16, 0) + 1
        want_out, want = run_oracle(case, {k: (None if v is None else v.float()) for k, v in t.items()}, torch.float32, intermediates=dtype)
        tol = 1e-2 if dtype == torch.float16 else 4e-2

        def cmp16():
            assert not all_finite(out, grads), "16-bit path: NaN from a read outside a tensor: %s" % all_finite(out, grads)
            assert_close("output", out.float(), want_out, tol)
            for k, v in grads.items():
                if v is not None and want[k] is not None:
                    assert_close(k, v.float(), want[k], tol)
        ok = check("%s %s %s" % (case["name"], dtype, {k: case[k] for k in ("op", "B", "C", "O", "in_sz", "k", "stride", "padding", "dilation", "groups", "dgroups")}), cmp16)
        n[2] += 1
        viol = run_guarded_outputs(case, t)
        if viol:
            print("FAIL %s %s guard margins of caller tensors: %s" % (case["name"], dtype, viol), flush=True)
            ok = False
        bad += 0 if ok else 1
        seed += 1
    print("fuzz_more: seeds %d..%d, %d fp32 shapes (matrix vs generic path), %d of them also vs the oracle, %d 16-bit shapes vs the oracle; "
          "%d FAILED; kernel paths %s; %.0f s" % (first, seed - 1, n[0], n[1], n[2], bad, paths, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
