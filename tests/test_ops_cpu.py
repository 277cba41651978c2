"""torch.library registration: schema, fake kernels, validation --
everything that can be checked without a GPU."""
import pytest
import torch

import modulated_deform_conv_amd.ops as ops


def _meta_args(nd=2, modulated=True, B=2, C=8, O=6, size=(9, 7, 5), k=3, stride=2, pad=1, dil=1,
               groups=2, dg=2, bias=True, dtype=torch.float32, device="meta"):
    size = tuple(size[:nd])
    osz = ops.output_size(size, (k,) * nd, (stride,) * nd, (pad,) * nd, (dil,) * nd)
    K = k ** nd
    e = lambda *s: torch.empty(*s, dtype=dtype, device=device)
    return dict(input=e(B, C, *size), offset=e(B, dg * nd * K, *osz),
                mask=e(B, dg * K, *osz) if modulated else None, weight=e(O, C // groups, *(k,) * nd),
                bias=e(O) if bias else None, stride=[stride] * nd, padding=[pad] * nd,
                dilation=[dil] * nd, groups=groups, deformable_groups=dg, in_step=64), osz


def test_ops_are_registered():
    assert hasattr(torch.ops.mdconv, "deform_conv")
    assert hasattr(torch.ops.mdconv, "deform_conv_backward")
    schema = str(torch.ops.mdconv.deform_conv.default._schema)
    assert "Tensor? mask" in schema and "Tensor? bias" in schema and "[] stride" in schema


@pytest.mark.parametrize("nd", [2, 3])
@pytest.mark.parametrize("modulated", [False, True])
@pytest.mark.parametrize("bias", [False, True])
def test_fake_kernels_infer_shapes(nd, modulated, bias):
    a, osz = _meta_args(nd=nd, modulated=modulated, bias=bias)
    out = ops.deform_conv(**a)
    assert out.device.type == "meta" and list(out.shape) == [2, 6] + osz
    gi, goff, gm, gw, gb = ops.deform_conv_backward(torch.empty_like(out), **a)
    assert gi.shape == a["input"].shape and goff.shape == a["offset"].shape
    assert gw.shape == a["weight"].shape
    assert gm.shape == (a["mask"].shape if modulated else (0,))
    assert gb.shape == (a["bias"].shape if bias else (0,))


def test_fake_tensor_mode_traces_forward_and_backward():
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        a, osz = _meta_args(device="cpu")
        for n in ("input", "offset", "mask", "weight", "bias"):
            a[n].requires_grad_(True)
        out = ops.deform_conv(**a)
        assert list(out.shape) == [2, 6] + osz
        out.sum().backward()
        assert a["weight"].grad.shape == a["weight"].shape
        assert a["mask"].grad.shape == a["mask"].shape


@pytest.mark.parametrize("breakage, match", [
    (lambda a: a.update(offset=a["offset"][:, :-1]), "offset must be"),
    (lambda a: a.update(mask=a["mask"][:, :, :-1]), "mask must be"),
    (lambda a: a.update(groups=3), "do not match weight"),
    (lambda a: a.update(deformable_groups=3), "divisible by deformable_groups"),
    (lambda a: a.update(in_step=0), "must be > 0"),
    (lambda a: a.update(stride=[1]), "stride must have"),
    (lambda a: a.update(bias=a["bias"][:-1]), "bias must be"),
    (lambda a: a.update(weight=a["weight"].double()), "one dtype"),
    (lambda a: a.update(padding=[0, 0], dilation=[9, 9]), "empty output"),
])
def test_validation_the_reference_lacks(breakage, match):
    a, _ = _meta_args()
    breakage(a)
    with pytest.raises(RuntimeError, match=match):
        ops.deform_conv(**a)


def test_cpu_tensors_are_refused():
    """No CPU path in the product (reference wrapper: NotImplementedError, :22-23)."""
    a, _ = _meta_args(device="cpu")
    with pytest.raises(NotImplementedError):
        ops.deform_conv(**a)
