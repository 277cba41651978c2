"""GPU tests of the autograd Functions / nn.Modules (the reference's L2/L3 surface) including the
my_test.py scenario (reference my_test.py:1-35, known answers in SURVEY.md section 4)."""
import pytest
import torch
import torch.nn.functional as F

from tests.cases import CASE_BY_NAME, make_inputs
from tests.util import assert_close, run_oracle

pytestmark = pytest.mark.gpu


def test_my_test_scenario_on_device():
    from modulated_deform_conv_amd.modulated_deform_conv import deform_conv2d, modulated_deform_conv2d
    counts = torch.tensor([[4, 6, 6, 6, 4], [6, 9, 9, 9, 6], [6, 9, 9, 9, 6], [6, 9, 9, 9, 6],
                           [4, 6, 6, 6, 4]], dtype=torch.float32, device="cuda")
    for fn, modulated in ((deform_conv2d, False), (modulated_deform_conv2d, True)):
        data = torch.ones(1, 1, 5, 5, device="cuda", requires_grad=True)
        offset = torch.zeros(1, 18, 5, 5, device="cuda", requires_grad=True)
        mask = torch.ones(1, 9, 5, 5, device="cuda", requires_grad=True)
        weight = torch.ones(1, 1, 3, 3, device="cuda", requires_grad=True)
        bias = torch.zeros(1, device="cuda", requires_grad=True)
        args = (weight, bias, 1, 1, 1, 1, 1, 64)
        out = fn(data, offset, mask, *args) if modulated else fn(data, offset, *args)
        assert torch.equal(out[0, 0], counts) and out.sum().item() == 169
        out.sum().backward()
        assert torch.equal(data.grad[0, 0], counts)
        assert torch.equal(weight.grad.flatten(),
                           torch.tensor([16., 20, 16, 20, 25, 20, 16, 20, 16], device="cuda"))
        assert bias.grad.item() == 25
        if modulated:
            gm0 = torch.zeros(5, 5, device="cuda")
            gm0[1:, 1:] = 1
            assert torch.equal(mask.grad[0, 0], gm0)
            assert offset.grad.abs().sum().item() == 52      # quirk Q2, modulated-2D flavour


@pytest.mark.parametrize("name", ["mdcn2d_s2_g4_dg2", "dcn2d_dil2_dg4", "dcn3d_basic", "mdcn3d_dil2_dg2"])
def test_module_forward_backward_matches_oracle(name):
    from modulated_deform_conv_amd import modulated_deform_conv as mdc
    case = CASE_BY_NAME[name]
    t = make_inputs(case, dtype=torch.float32, device="cuda")
    cls = {0: mdc.DeformConv2d, 1: mdc.ModulatedDeformConv2d, 2: mdc.DeformConv3d,
           3: mdc.ModulatedDeformConv3d}[case["op"]]
    mod = cls(case["C"], case["O"], case["k"], case["stride"], case["padding"], case["dilation"],
              case["groups"], case["dgroups"], bias=case["bias"], in_step=case["in_step"]).cuda()
    with torch.no_grad():
        mod.weight.copy_(t["weight"])
        if case["bias"]:
            mod.bias.copy_(t["bias"])
    x = t["input"].clone().requires_grad_()
    off = t["offset"].clone().requires_grad_()
    ins = [x, off]
    if t["mask"] is not None:
        ins.append(t["mask"].clone().requires_grad_())
    out = mod(*ins)
    out.backward(t["grad_output"])
    want_out, want = run_oracle(case, t, torch.float32)
    assert_close("output", out, want_out, 1e-4)
    assert_close("grad_input", x.grad, want["grad_input"], 1e-4)
    assert_close("grad_offset", off.grad, want["grad_offset"], 1e-4)
    assert_close("grad_weight", mod.weight.grad, want["grad_weight"], 1e-4)
    if t["mask"] is not None:
        assert_close("grad_mask", ins[2].grad, want["grad_mask"], 1e-4)
    if case["bias"]:
        assert_close("grad_bias", mod.bias.grad, want["grad_bias"], 1e-4)


def test_zero_offset_equals_conv2d_on_device():
    from modulated_deform_conv_amd.modulated_deform_conv import modulated_deform_conv2d
    torch.manual_seed(0)
    x = torch.randn(2, 16, 14, 14, device="cuda", requires_grad=True)
    w = torch.randn(32, 16, 3, 3, device="cuda", requires_grad=True)
    off = torch.zeros(2, 18, 14, 14, device="cuda")
    m = torch.ones(2, 9, 14, 14, device="cuda")
    out = modulated_deform_conv2d(x, off, m, w, None, 1, 1, 1, 1, 1, 64)
    ref = F.conv2d(x, w, None, 1, 1)
    assert_close("conv2d", out, ref, 1e-4)


def test_pack_module_runs():
    from modulated_deform_conv_amd import modulated_deform_conv as mdc
    torch.manual_seed(0)
    m = mdc.ModulatedDeformConv2dPack(8, 8, 3, padding=1, deformable_groups=2, bias=True).cuda()
    x = torch.randn(2, 8, 10, 10, device="cuda", requires_grad=True)
    y = m(x)
    y.square().mean().backward()
    assert y.shape == (2, 8, 10, 10) and torch.isfinite(x.grad).all()
    assert m.conv_offset.weight.grad is not None and m.conv_mask.weight.grad is not None


@pytest.mark.parametrize("nd", [2, 3])
def test_pack_fused_side_conv_equals_two_convs(nd):
    """The Pack modules run conv_offset and conv_mask as ONE convolution; output and every
    parameter gradient must equal the reference formulation with two (reference :755-785)."""
    from modulated_deform_conv_amd import modulated_deform_conv as mdc
    torch.manual_seed(1)
    cls = mdc.ModulatedDeformConv2dPack if nd == 2 else mdc.ModulatedDeformConv3dPack
    base = mdc.ModulatedDeformConv2d if nd == 2 else mdc.ModulatedDeformConv3d
    m = cls(8, 6, 3, stride=1, padding=1, deformable_groups=2, bias=True).cuda()
    assert set(k.split(".")[0] for k in m.state_dict()) == {"weight", "bias", "conv_offset", "conv_mask"}
    x = torch.randn(2, 8, *([7] * nd), device="cuda", requires_grad=True)
    y = m(x)
    y.square().sum().backward()
    got = [x.grad.clone()] + [p.grad.clone() for p in m.parameters()]
    x.grad = None
    m.zero_grad()
    y2 = base.forward(m, x, m.conv_offset(x), m.conv_mask(x))
    y2.square().sum().backward()
    want = [x.grad] + [p.grad for p in m.parameters()]
    assert_close("output", y, y2, 1e-5)
    for a, b in zip(got, want):
        assert_close("grad", a, b, 1e-4)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_autocast_runs_on_the_native_16bit_kernels(dt):
    """AMP integration: under autocast the op casts its inputs to the autocast dtype and runs the
    native fp16 / bf16 kernels (fp32 coordinates and accumulation inside); fp32 leaf tensors get
    fp32 gradients."""
    from modulated_deform_conv_amd import modulated_deform_conv as mdc, _capi
    torch.manual_seed(0)
    m = mdc.ModulatedDeformConv2d(32, 32, 3, padding=1, bias=True).cuda()
    x = torch.randn(2, 32, 9, 9, device="cuda", requires_grad=True)
    off = torch.randn(2, 18, 9, 9, device="cuda")
    mask = torch.rand(2, 9, 9, 9, device="cuda")
    ref = m(x, off, mask)
    with torch.autocast("cuda", dtype=dt):
        y = m(x, off.half(), mask)        # mixed input dtypes are fine under autocast
    assert y.dtype == dt and _capi.last_kernels() == "hp"
    assert_close("autocast", y.float(), ref, 2e-2 if dt == torch.float16 else 6e-2)   # inputs are rounded to 16 bits here and not in `ref`: not an oracle-parity tolerance
    y.float().sum().backward()
    assert x.grad is not None and x.grad.dtype == torch.float32 and torch.isfinite(x.grad).all()
    assert m.weight.grad.dtype == torch.float32


def test_pack_calls_hooked_side_convs_like_the_reference():
    """A hook / parametrisation on conv_offset or conv_mask must take effect (the reference CALLS
    the modules, modulated_deform_conv.py:779-783): the fused single-convolution path is only used
    for plain convolutions."""
    from modulated_deform_conv_amd import modulated_deform_conv as mdc
    torch.manual_seed(2)
    m = mdc.ModulatedDeformConv2dPack(8, 8, 3, padding=1, bias=True).cuda()
    x = torch.randn(2, 8, 9, 9, device="cuda")
    y_plain = m(x)
    calls = []
    h = m.conv_mask.register_forward_hook(lambda mod, inp, out: (calls.append(1), out * 0)[1])
    y_hooked = m(x)
    h.remove()
    assert calls and not torch.allclose(y_plain, y_hooked)
    assert_close("hook removed", m(x), y_plain, 1e-6)
    torch.nn.utils.parametrizations.weight_norm(m.conv_offset)
    y_wn = m(x)     # parametrised module: the path that calls the modules
    y_ref = mdc.ModulatedDeformConv2d.forward(m, x, m.conv_offset(x), m.conv_mask(x))
    assert_close("weight_norm", y_wn, y_ref, 1e-6)


def test_overlapped_reduce_after_autograd_backward():
    """The weights-ready event is keyed by (device, stream), not by host thread: it is found after
    loss.backward(), which runs the op on an autograd worker thread (world size 1: the reducer
    takes its CPU / single-process path only for the collective itself)."""
    from modulated_deform_conv_amd import modulated_deform_conv as mdc, _capi
    torch.manual_seed(3)
    m = mdc.ModulatedDeformConv2d(64, 64, 3, padding=1, bias=True).cuda()
    x = torch.randn(4, 64, 28, 28, device="cuda", requires_grad=True)
    off = torch.randn(4, 18, 28, 28, device="cuda")
    mask = torch.rand(4, 9, 28, 28, device="cuda")
    m(x, off, mask).square().sum().backward()
    ref_w, ref_b = m.weight.grad.clone(), m.bias.grad.clone()
    side = torch.cuda.Stream()
    for _ in range(3):
        m.zero_grad()
        m(x, off, mask).square().sum().backward()
        _capi.stream_wait_weight_ready(side, producer=torch.cuda.current_stream())
        with torch.cuda.stream(side):
            gw, gb = m.weight.grad.clone(), m.bias.grad.clone()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert torch.equal(gw, ref_w) and torch.equal(gb, ref_b)
    with pytest.raises(RuntimeError, match="no backward"):
        _capi.stream_wait_weight_ready(side, producer=torch.cuda.Stream())


def test_weight_ready_event_orders_a_side_stream():
    """mdconv_stream_wait_weight_ready: a side stream that waits for the event sees the final
    grad_weight / grad_bias of the backward just issued (the hook of the overlapped all-reduce)."""
    from modulated_deform_conv_amd import MDCONV_CUDA as M, _capi
    case = CASE_BY_NAME["cfg2s_mdcn2d_c64_28x28_b4"]
    t = make_inputs(case, device="cuda")
    geo = (3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 64, True)
    side = torch.cuda.Stream()
    ref = M.modulated_deform_conv2d_backward_cuda(t["input"], t["weight"], t["bias"], t["offset"],
                                                  t["mask"], t["grad_output"], *geo)
    torch.cuda.synchronize()
    for _ in range(5):
        gi, goff, gm, gw, gb = M.modulated_deform_conv2d_backward_cuda(
            t["input"], t["weight"], t["bias"], t["offset"], t["mask"], t["grad_output"], *geo)
        _capi.stream_wait_weight_ready(side)
        with torch.cuda.stream(side):
            gw2, gb2 = gw.clone(), gb.clone()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert torch.equal(gw2, ref[3]) and torch.equal(gb2, ref[4])
        # grad_input is only reproducible to rounding: the order of a pixel's CSR list depends on
        # integer-atomic arrival order (the reference's float atomics are no better)
        assert_close("grad_input", gi, ref[0], 1e-5)


@pytest.mark.parametrize("name, dtype", [
    ("cfg2s_mdcn2d_c64_28x28_b4", torch.float32), ("cfg4s_dcn3d_c16_12cubed_b2", torch.float32),
    ("mfma_mdcn2d_g4_dg2_c128_o64", torch.float32),
    ("mfma_split_dcn2d_g2_dg4_c128_o64", torch.float32),   # per-deformable-group slices: strided copy nodes
    ("cfg5s_mdcn3d_c16_dil2", torch.float16),       # hp_bwd3 + hp_gemm2 + two-pass gather, tails forked
    ("mfma_mdcn2d_g4_dg2_c128_o64", torch.float16)])   # hp_bwd2 (fused)
def test_hip_graph_capture_and_replay(name, dtype):
    """Forward + backward are a plain sequence of kernel launches on the caller's stream and the library's
    forked side stream (event fork / join, no memset nodes, no host synchronisation), so they capture into a
    HIP graph; the replay must reproduce the eager results."""
    from tests.util import run_product
    case = CASE_BY_NAME[name]
    t = make_inputs(case, dtype=dtype, device="cuda")
    ref_out, ref_g, paths = run_product(case, t, "auto")
    assert paths[1] == "mfma"
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        run_product(case, t, "auto")
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out, g, _ = run_product(case, t, "auto")
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    tol = 1e-5 if dtype == torch.float32 else 2e-3   # grad_input: list order depends on integer-atomic arrival
    assert_close("output", out.float(), ref_out.float(), 1e-6)
    for k, v in g.items():
        if v is not None and ref_g[k] is not None:
            assert_close(k, v.float(), ref_g[k].float(), tol, 4 * tol)


@pytest.mark.parametrize("cls_name, nd, modulated, kw", [
    ("ModulatedDeformConv2dPack", 2, True, dict(stride=2, deformable_groups=2)),
    ("DeformConv2dPack", 2, False, dict(groups=2)),
    ("ModulatedDeformConv3dPack", 3, True, dict()),
    ("DeformConv3dPack", 3, False, dict(deformable_groups=2)),
])
def test_pack_module_matches_oracle(cls_name, nd, modulated, kw):
    """*Pack modules against the ORACLE (reference modulated_deform_conv.py:730-839): offset (and
    mask) are what the side convolutions produce -- same kernel / stride / padding as the main
    convolution, dilation NOT forwarded, no sigmoid on the mask (the reference's quirks) -- and the
    oracle runs the deformable convolution on them; the gradients of x and of the side convolutions'
    parameters are chained through F.conv on the CPU."""
    import oracle
    from modulated_deform_conv_amd import modulated_deform_conv as mdc
    torch.manual_seed(7)
    C, O, B = 16, 24, 2
    sp = (10, 9) if nd == 2 else (5, 6, 5)
    stride, groups, dg = kw.get("stride", 1), kw.get("groups", 1), kw.get("deformable_groups", 1)
    mod = getattr(mdc, cls_name)(C, O, 3, stride=stride, padding=1, groups=groups,
                                 deformable_groups=dg, bias=True).cuda()
    with torch.no_grad():
        mod.bias.normal_(0, 0.1)
        mod.conv_offset.bias.normal_(0, 0.3)       # the reference initialises these to 0; make them count
        mod.conv_offset.weight.mul_(6.0)           # offsets of about a pixel
        if modulated:
            mod.conv_mask.bias.normal_(0.5, 0.2)
    x = torch.randn(B, C, *sp, device="cuda", requires_grad=True)
    out = mod(x)
    go = torch.randn_like(out)
    out.backward(go)
    # expected: side convolutions on the CPU (autograd), deformable convolution by the oracle
    conv = F.conv2d if nd == 2 else F.conv3d
    xc = x.detach().cpu().requires_grad_()
    cw = mod.conv_offset.weight.detach().cpu().requires_grad_()
    cb = mod.conv_offset.bias.detach().cpu().requires_grad_()
    off = conv(xc, cw, cb, stride, 1)              # dilation not forwarded
    if modulated:
        mw = mod.conv_mask.weight.detach().cpu().requires_grad_()
        mb = mod.conv_mask.bias.detach().cpu().requires_grad_()
        mask = conv(xc, mw, mb, stride, 1)         # no sigmoid
    op = {(2, False): oracle.DCN2D, (2, True): oracle.MDCN2D, (3, False): oracle.DCN3D, (3, True): oracle.MDCN3D}[(nd, modulated)]
    w, b = mod.weight.detach().cpu(), mod.bias.detach().cpu()
    md = mask.detach() if modulated else None
    want_out = oracle.forward(op, xc.detach(), w, b, off.detach(), md, stride, 1, 1, groups, dg, 64, dtype=torch.float32)
    g = oracle.backward(op, xc.detach(), w, b, off.detach(), md, go.cpu(), stride, 1, 1, groups, dg, 64, dtype=torch.float32)
    heads, grads = [off], [g["grad_offset"]]
    if modulated:
        heads.append(mask); grads.append(g["grad_mask"])
    torch.autograd.backward(heads, grads)
    assert_close("output", out, want_out, 1e-4)
    assert_close("x.grad", x.grad, g["grad_input"] + xc.grad, 1e-4)
    assert_close("weight.grad", mod.weight.grad, g["grad_weight"], 1e-4)
    assert_close("bias.grad", mod.bias.grad, g["grad_bias"], 1e-4)
    assert_close("conv_offset.weight.grad", mod.conv_offset.weight.grad, cw.grad, 1e-4)
    assert_close("conv_offset.bias.grad", mod.conv_offset.bias.grad, cb.grad, 1e-4)
    if modulated:
        assert_close("conv_mask.weight.grad", mod.conv_mask.weight.grad, mw.grad, 1e-4)
        assert_close("conv_mask.bias.grad", mod.conv_mask.bias.grad, mb.grad, 1e-4)
