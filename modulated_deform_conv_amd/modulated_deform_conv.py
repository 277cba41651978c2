"""Operator surface of the reference's ``modulated_deform_conv.py`` on the MI355X-native backend.

Same public names, constructor / call signatures, parameter names and shapes, and autograd
tuple arity as the reference (modulated_deform_conv.py:9-352 Functions, :354-537 Modules,
:730-839 Pack modules), so user code and checkpoints (state_dict keys ``weight``, ``bias``,
``conv_offset.*``, ``conv_mask.*``) interchange.  Written once for N spatial dims instead of
four times; all compute goes through ``MDCONV_CUDA`` (this package's ctypes binding of
libmdconv_hip.so).  CPU tensors raise ``NotImplementedError`` exactly like the reference
(modulated_deform_conv.py:22-23): there is no CPU fallback in the product.
"""
import math

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair, _triple

from . import MDCONV_CUDA, _capi
from .distributed import fused_grad_buffers

__all__ = [
    "DeformConv2dFunction", "ModulatedDeformConv2dFunction", "DeformConv3dFunction",
    "ModulatedDeformConv3dFunction", "deform_conv2d", "modulated_deform_conv2d", "deform_conv3d",
    "modulated_deform_conv3d", "DeformConv2d", "ModulatedDeformConv2d", "DeformConv3d",
    "ModulatedDeformConv3d", "DeformConv2dPack", "ModulatedDeformConv2dPack", "DeformConv3dPack",
    "ModulatedDeformConv3dPack",
]


def _ntuple(nd):
    return _pair if nd == 2 else _triple


def _output_shape(input, weight, stride, padding, dilation):
    """(n + 2p - (d(k-1)+1)) // s + 1 per axis -- reference _infer_shape, :84-91."""
    spatial = tuple((n + 2 * p - (d * (k - 1) + 1)) // s + 1
                    for n, k, s, p, d in zip(input.shape[2:], weight.shape[2:], stride, padding, dilation))
    return (input.size(0), weight.size(0)) + spatial


def _make_function(nd, modulated, name):
    """Build one autograd.Function of the family.

    forward(ctx, input, offset, [mask,] weight, bias=None, stride=1, padding=0, dilation=1,
            groups=1, deformable_groups=1, in_step=64)
    backward -> (grad_input, grad_offset, [grad_mask,] grad_weight, grad_bias|None, None x 6)
    """
    tup = _ntuple(nd)
    fwd = getattr(MDCONV_CUDA, ("modulated_" if modulated else "") + "deform_conv%dd_forward_cuda" % nd)
    bwd = getattr(MDCONV_CUDA, ("modulated_" if modulated else "") + "deform_conv%dd_backward_cuda" % nd)
    returns_tensors = modulated and nd == 2   # the one pair whose reference ABI allocates its results

    def _setup(ctx, input, bias, stride, padding, dilation, groups, deformable_groups, in_step):
        ctx.stride, ctx.padding, ctx.dilation = tup(stride), tup(padding), tup(dilation)
        ctx.groups, ctx.deformable_groups, ctx.in_step = groups, deformable_groups, in_step
        ctx.with_bias = bias is not None
        if not ctx.with_bias:
            bias = input.new_empty(0)   # the reference's "fake tensor", :19-21
        if not input.is_cuda:
            raise NotImplementedError
        return bias

    def _geometry(ctx, weight):
        return tuple(weight.shape[2:]) + ctx.stride + ctx.padding + ctx.dilation + \
            (ctx.groups, ctx.deformable_groups, ctx.in_step, ctx.with_bias)

    def _forward(ctx, input, offset, mask, weight, bias):
        needs_grad = weight.requires_grad or offset.requires_grad or input.requires_grad or \
            (modulated and mask.requires_grad)
        # gradients go back in the dtypes the caller's tensors have (the autocast cast below happens
        # inside this Function, so autograd does not see it)
        ctx.in_dtypes = tuple(None if t is None else t.dtype for t in (input, offset, mask, weight, bias))
        if torch.is_autocast_enabled("cuda"):
            # AMP: run in the autocast dtype on the native 16-bit kernels
            # (fp16 / bf16 operands, fp32 coordinates, interpolation weights and accumulators)
            dt = torch.get_autocast_dtype("cuda")
            cast = lambda t: t if t is None or not t.is_floating_point() or t.dtype == dt else t.to(dt)
            input, offset, mask, weight, bias = (cast(t) for t in (input, offset, mask, weight, bias))
        if needs_grad:
            saved = (input, offset, mask, weight, bias) if modulated else (input, offset, weight, bias)
            ctx.save_for_backward(*saved)
        geo = _geometry(ctx, weight)
        if returns_tensors:
            return fwd(input, weight, bias, offset, mask, *geo)
        output = input.new_empty(_output_shape(input, weight, ctx.stride, ctx.padding, ctx.dilation))
        if modulated:
            fwd(input, weight, bias, offset, mask, output, *geo)
        else:
            fwd(input, weight, bias, offset, output, *geo)
        return output

    def _backward(ctx, grad_output):
        grad_output = grad_output.contiguous()
        if not grad_output.is_cuda:
            raise NotImplementedError
        if modulated:
            input, offset, mask, weight, bias = ctx.saved_tensors
        else:
            input, offset, weight, bias = ctx.saved_tensors
            mask = None
        geo = _geometry(ctx, weight)
        if returns_tensors:
            grad_input, grad_offset, grad_mask, grad_weight, grad_bias = bwd(
                input, weight, bias, offset, mask, grad_output, *geo)
        else:
            # the reference wrapper zero-fills and the entry points add (:53-56); here the buffers
            # are fresh, so the library is asked to write them instead (mdconv_set_accumulate)
            grad_input = torch.empty_like(input, memory_format=torch.contiguous_format)
            grad_offset = torch.empty_like(offset)
            grad_weight, grad_bias = fused_grad_buffers(weight, bias)   # one flat buffer: one in-place all-reduce (distributed.py)
            with _capi.overwrite_grads():
                if modulated:
                    grad_mask = torch.empty_like(mask)
                    bwd(input, weight, bias, offset, mask, grad_input, grad_weight, grad_bias,
                        grad_offset, grad_mask, grad_output, *geo)
                else:
                    grad_mask = None
                    bwd(input, weight, bias, offset, grad_input, grad_weight, grad_bias, grad_offset,
                        grad_output, *geo)
        if not ctx.with_bias:
            grad_bias = None
        back = lambda g, dt: g if g is None or dt is None or g.dtype == dt else g.to(dt)
        grad_input, grad_offset, grad_mask, grad_weight, grad_bias = (
            back(g, dt) for g, dt in zip((grad_input, grad_offset, grad_mask, grad_weight, grad_bias), ctx.in_dtypes))
        head = (grad_input, grad_offset, grad_mask) if modulated else (grad_input, grad_offset)
        return head + (grad_weight, grad_bias) + (None,) * 6

    if modulated:
        def forward(ctx, input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1,
                    groups=1, deformable_groups=1, in_step=64):
            bias = _setup(ctx, input, bias, stride, padding, dilation, groups, deformable_groups, in_step)
            return _forward(ctx, input, offset, mask, weight, bias)
    else:
        def forward(ctx, input, offset, weight, bias=None, stride=1, padding=0, dilation=1,
                    groups=1, deformable_groups=1, in_step=64):
            bias = _setup(ctx, input, bias, stride, padding, dilation, groups, deformable_groups, in_step)
            return _forward(ctx, input, offset, None, weight, bias)

    def _infer_shape(ctx, input, weight):
        return _output_shape(input, weight, ctx.stride, ctx.padding, ctx.dilation)

    return type(name, (Function,), {
        # AMP: custom_fwd records the autocast state (custom_bwd replays it in backward); the cast to
        # the autocast dtype happens in _forward so that fp16 AND bf16 autocast both work
        "forward": staticmethod(custom_fwd(forward, device_type="cuda")),
        "backward": staticmethod(custom_bwd(once_differentiable(_backward), device_type="cuda")),
        "_infer_shape": staticmethod(_infer_shape),
        "__doc__": "%s-D %sdeformable convolution (reference modulated_deform_conv.py)." % (
            nd, "modulated " if modulated else ""),
    })


DeformConv2dFunction = _make_function(2, False, "DeformConv2dFunction")
ModulatedDeformConv2dFunction = _make_function(2, True, "ModulatedDeformConv2dFunction")
DeformConv3dFunction = _make_function(3, False, "DeformConv3dFunction")
ModulatedDeformConv3dFunction = _make_function(3, True, "ModulatedDeformConv3dFunction")

deform_conv2d = DeformConv2dFunction.apply
modulated_deform_conv2d = ModulatedDeformConv2dFunction.apply
deform_conv3d = DeformConv3dFunction.apply
modulated_deform_conv3d = ModulatedDeformConv3dFunction.apply


class _DeformConvNd(nn.Module):
    """Shared body of the four modules (reference :354-537).  ``bias`` defaults to False."""
    _nd = 2
    _modulated = False
    _op = None

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, deformable_groups=1, bias=False, in_step=64):
        super().__init__()
        assert in_channels % groups == 0, \
            'in_channels {} cannot be divisible by groups {}'.format(in_channels, groups)
        assert out_channels % groups == 0, \
            'out_channels {} cannot be divisible by groups {}'.format(out_channels, groups)
        tup = _ntuple(self._nd)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = tup(kernel_size), tup(stride)
        self.padding, self.dilation = tup(padding), tup(dilation)
        self.groups, self.deformable_groups, self.in_step = groups, deformable_groups, in_step
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *self.kernel_size))
        self.with_bias = bias
        self.bias = nn.Parameter(torch.Tensor(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1. / math.sqrt(self.in_channels * math.prod(self.kernel_size))
        self.weight.data.uniform_(-stdv, stdv)
        if self.with_bias:
            self.bias.data.fill_(0)

    def _conv_args(self):
        return (self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups,
                self.deformable_groups, self.in_step)

    def extra_repr(self):
        return ("{in_channels}, {out_channels}, kernel_size={kernel_size}, stride={stride}, "
                "padding={padding}, dilation={dilation}, groups={groups}, "
                "deformable_groups={deformable_groups}, bias={with_bias}").format(**self.__dict__)


class DeformConv2d(_DeformConvNd):
    _nd, _modulated = 2, False

    def forward(self, x, offset):
        return deform_conv2d(x, offset, *self._conv_args())


class ModulatedDeformConv2d(_DeformConvNd):
    _nd, _modulated = 2, True

    def forward(self, x, offset, mask):
        return modulated_deform_conv2d(x, offset, mask, *self._conv_args())


class DeformConv3d(_DeformConvNd):
    _nd, _modulated = 3, False

    def forward(self, x, offset):
        return deform_conv3d(x, offset, *self._conv_args())


class ModulatedDeformConv3d(_DeformConvNd):
    _nd, _modulated = 3, True

    def forward(self, x, offset, mask):
        return modulated_deform_conv3d(x, offset, mask, *self._conv_args())


class _PackMixin:
    """Adds the offset (and mask) producing convolutions (reference :730-839).

    Kept quirks: ``dilation`` is not forwarded to ``conv_offset`` / ``conv_mask``, no sigmoid is
    applied to the mask, and both convs are initialised U(+-1/sqrt(C_in*prod(k))) with zero bias.
    """

    def _make_side_convs(self):
        conv = nn.Conv2d if self._nd == 2 else nn.Conv3d
        K = math.prod(self.kernel_size)
        self.conv_offset = conv(self.in_channels, self.deformable_groups * self._nd * K,
                                kernel_size=self.kernel_size, stride=self.stride,
                                padding=self.padding, bias=True)
        if self._modulated:
            self.conv_mask = conv(self.in_channels, self.deformable_groups * K,
                                  kernel_size=self.kernel_size, stride=self.stride,
                                  padding=self.padding, bias=True)
        self.init_offset()

    def init_offset(self):
        """(Re-)initialise the side convolutions (reference init_offset / init_offset_mask)."""
        stdv = 1. / math.sqrt(self.in_channels * math.prod(self.kernel_size))
        for m in (self.conv_offset, getattr(self, "conv_mask", None)):
            if m is not None:
                m.weight.data.uniform_(-stdv, stdv)
                m.bias.data.zero_()

    init_offset_mask = init_offset


def _make_pack(base, name):
    def __init__(self, *args, **kwargs):
        base.__init__(self, *args, **kwargs)
        self._make_side_convs()

    def _plain(m, conv_cls):
        """True if calling F.conv directly on m's parameters is equivalent to calling m: an
        unmodified nn.ConvNd without hooks (weight_norm / spectral_norm / pruning recompute
        `.weight` in a pre-forward hook; quantisation or LoRA wrappers replace the module)."""
        return (type(m) is conv_cls and not m._forward_hooks and not m._forward_pre_hooks
                and not m._backward_hooks and not getattr(m, "_backward_pre_hooks", None)
                and m.padding_mode == "zeros" and m.groups == 1 and m.bias is not None
                and not torch.nn.modules.module._global_forward_hooks
                and not torch.nn.modules.module._global_forward_pre_hooks)

    if base._modulated:
        def forward(self, x):
            # ONE side convolution for offset and mask: the two parameter
            # sets stay separate modules (state_dict keys conv_offset.* / conv_mask.* as in the
            # reference) and are concatenated along the output channels, so the input is read once
            # and one library launch replaces two.  Same numbers as two convolutions.  Only when
            # both are plain convolutions of identical geometry; otherwise the modules are CALLED,
            # exactly like the reference (:779-783), so hooks / parametrisations / replaced
            # submodules keep working.
            co, cm = self.conv_offset, self.conv_mask
            conv_cls = nn.Conv2d if self._nd == 2 else nn.Conv3d
            if (_plain(co, conv_cls) and _plain(cm, conv_cls) and co.stride == cm.stride
                    and co.padding == cm.padding and co.dilation == cm.dilation
                    and co.kernel_size == cm.kernel_size and co.weight.dtype == cm.weight.dtype):
                conv = torch.nn.functional.conv2d if self._nd == 2 else torch.nn.functional.conv3d
                y = conv(x, torch.cat((co.weight, cm.weight)), torch.cat((co.bias, cm.bias)),
                         co.stride, co.padding, co.dilation)
                n_off = co.out_channels
                return base.forward(self, x, y[:, :n_off].contiguous(), y[:, n_off:].contiguous())
            return base.forward(self, x, co(x), cm(x))
    else:
        def forward(self, x):
            return base.forward(self, x, self.conv_offset(x))
    return type(name, (_PackMixin, base), {"__init__": __init__, "forward": forward})


DeformConv2dPack = _make_pack(DeformConv2d, "DeformConv2dPack")
ModulatedDeformConv2dPack = _make_pack(ModulatedDeformConv2d, "ModulatedDeformConv2dPack")
DeformConv3dPack = _make_pack(DeformConv3d, "DeformConv3dPack")
ModulatedDeformConv3dPack = _make_pack(ModulatedDeformConv3d, "ModulatedDeformConv3dPack")
