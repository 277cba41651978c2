"""``torch.library`` registration of the deformable-convolution path.

Two dispatcher-visible operators, written once for 2-D / 3-D and plain / modulated:

    mdconv::deform_conv(input, offset, mask?, weight, bias?, stride, padding, dilation,
                        groups, deformable_groups, in_step) -> Tensor
    mdconv::deform_conv_backward(grad_output, input, offset, mask?, weight, bias?, ...)
        -> (grad_input, grad_offset, grad_mask, grad_weight, grad_bias)

with fake (meta) kernels, so FakeTensor / ``torch.compile`` / ``torch.export`` can trace through
them without running a kernel, and an autograd formula that links the two.  The CUDA(=HIP)
implementations call the same eight ``MDCONV_CUDA`` entry points as the legacy
``autograd.Function`` classes (reference call sites modulated_deform_conv.py:28, 57, 112, 142,
194, 225, 281, 313) -- the eight positional exports stay the drop-in boundary; this is the
modern front door for new callers.

Unlike the reference boundary (mdeformable_conv.cu:127-148 checks only contiguity, kernel dims
and C_in = weight.size(1) * group), ``_validate`` checks every shape, dtype and device
relation the kernels rely on and raises ``RuntimeError`` with the offending sizes.
There is no CPU implementation: calling the op on CPU tensors raises ``NotImplementedError``
from the dispatcher, like the reference wrapper (modulated_deform_conv.py:22-23).
"""
from typing import List, Optional, Tuple

import torch

from . import MDCONV_CUDA, _capi

__all__ = ["deform_conv", "deform_conv_backward", "output_size"]


def output_size(in_size, kernel, stride, padding, dilation):
    """(n + 2p - (d(k-1)+1)) // s + 1 per axis (mdeformable_conv.cu:150-153)."""
    return [(n + 2 * p - (d * (k - 1) + 1)) // s + 1
            for n, k, s, p, d in zip(in_size, kernel, stride, padding, dilation)]


def _validate(input, offset, mask, weight, bias, stride, padding, dilation, groups,
              deformable_groups, in_step, grad_output=None):
    nd = input.dim() - 2
    if nd not in (2, 3):
        raise RuntimeError("deform_conv: input must be [B, C, H, W] or [B, C, H, W, L], got %s"
                           % (tuple(input.shape),))
    for name, v in (("stride", stride), ("padding", padding), ("dilation", dilation)):
        if len(v) != nd:
            raise RuntimeError("deform_conv: %s must have %d entries, got %s" % (name, nd, list(v)))
    if any(s <= 0 for s in stride) or any(d <= 0 for d in dilation) or any(p < 0 for p in padding):
        raise RuntimeError("deform_conv: stride/dilation must be > 0 and padding >= 0")
    if weight.dim() != nd + 2:
        raise RuntimeError("deform_conv: weight must have %d dims, got %s" % (nd + 2, tuple(weight.shape)))
    B, C = input.shape[0], input.shape[1]
    O = weight.shape[0]
    if groups <= 0 or deformable_groups <= 0 or in_step <= 0:
        raise RuntimeError("deform_conv: groups, deformable_groups and in_step must be > 0")
    if C != weight.shape[1] * groups or O % groups:
        raise RuntimeError("deform_conv: C_in=%d, C_out=%d do not match weight %s with groups=%d"
                           % (C, O, tuple(weight.shape), groups))
    if C % deformable_groups:
        raise RuntimeError("deform_conv: C_in=%d is not divisible by deformable_groups=%d"
                           % (C, deformable_groups))
    kernel = list(weight.shape[2:])
    osz = output_size(input.shape[2:], kernel, stride, padding, dilation)
    if any(o <= 0 for o in osz):
        raise RuntimeError("deform_conv: empty output %s" % (osz,))
    K = 1
    for k in kernel:
        K *= k
    want_off = [B, deformable_groups * nd * K] + osz
    if list(offset.shape) != want_off:
        raise RuntimeError("deform_conv: offset must be %s, got %s" % (want_off, list(offset.shape)))
    if mask is not None and list(mask.shape) != [B, deformable_groups * K] + osz:
        raise RuntimeError("deform_conv: mask must be %s, got %s"
                           % ([B, deformable_groups * K] + osz, list(mask.shape)))
    if bias is not None and list(bias.shape) != [O]:
        raise RuntimeError("deform_conv: bias must be [%d], got %s" % (O, list(bias.shape)))
    if grad_output is not None and list(grad_output.shape) != [B, O] + osz:
        raise RuntimeError("deform_conv: grad_output must be %s, got %s"
                           % ([B, O] + osz, list(grad_output.shape)))
    tensors = [t for t in (input, offset, mask, weight, bias, grad_output) if t is not None]
    if any(t.dtype != input.dtype for t in tensors):
        raise RuntimeError("deform_conv: all tensors must share one dtype, got %s"
                           % [str(t.dtype) for t in tensors])
    if input.dtype not in (torch.float32, torch.float16, torch.float64, torch.bfloat16):   # = MDCONV_CUDA._DTYPES
        raise RuntimeError("deform_conv: dtype must be float32 / float16 / bfloat16 / float64, got %s" % input.dtype)
    if any(t.device != input.device for t in tensors):
        raise RuntimeError("deform_conv: all tensors must be on one device")
    return nd, osz


def _geometry(weight, stride, padding, dilation, groups, deformable_groups, in_step, with_bias):
    return tuple(weight.shape[2:]) + tuple(stride) + tuple(padding) + tuple(dilation) + \
        (groups, deformable_groups, in_step, with_bias)


def _input_layout(input):
    """A dense channels-last 16-bit `input` goes to the entry points as it is: they gather from that layout where the
    kernels of the direction take it and make their own contiguous copy where not (MDCONV_CUDA._layout)."""
    return input if MDCONV_CUDA._is_channels_last(input) else input.contiguous()


def _entry(nd, modulated, backward):
    return getattr(MDCONV_CUDA, "%sdeform_conv%dd_%s_cuda"
                   % ("modulated_" if modulated else "", nd, "backward" if backward else "forward"))


@torch.library.custom_op("mdconv::deform_conv", mutates_args=(), device_types="cuda")
def deform_conv(input: torch.Tensor, offset: torch.Tensor, mask: Optional[torch.Tensor],
                weight: torch.Tensor, bias: Optional[torch.Tensor], stride: List[int],
                padding: List[int], dilation: List[int], groups: int, deformable_groups: int,
                in_step: int) -> torch.Tensor:
    nd, osz = _validate(input, offset, mask, weight, bias, stride, padding, dilation, groups,
                        deformable_groups, in_step)
    input, offset, weight = _input_layout(input), offset.contiguous(), weight.contiguous()
    mask = None if mask is None else mask.contiguous()
    b = input.new_empty(0) if bias is None else bias.contiguous()
    geo = _geometry(weight, stride, padding, dilation, groups, deformable_groups, in_step,
                    bias is not None)
    fn = _entry(nd, mask is not None, False)
    if mask is not None and nd == 2:   # the one export that allocates its result
        return fn(input, weight, b, offset, mask, *geo)
    out = torch.empty([input.shape[0], weight.shape[0]] + osz, dtype=input.dtype, device=input.device)
    if mask is not None:
        fn(input, weight, b, offset, mask, out, *geo)
    else:
        fn(input, weight, b, offset, out, *geo)
    return out


@deform_conv.register_fake
def _(input, offset, mask, weight, bias, stride, padding, dilation, groups, deformable_groups,
      in_step):
    _, osz = _validate(input, offset, mask, weight, bias, stride, padding, dilation, groups,
                       deformable_groups, in_step)
    return input.new_empty([input.shape[0], weight.shape[0]] + osz)


@torch.library.custom_op("mdconv::deform_conv_backward", mutates_args=(), device_types="cuda")
def deform_conv_backward(grad_output: torch.Tensor, input: torch.Tensor, offset: torch.Tensor,
                         mask: Optional[torch.Tensor], weight: torch.Tensor,
                         bias: Optional[torch.Tensor], stride: List[int], padding: List[int],
                         dilation: List[int], groups: int, deformable_groups: int,
                         in_step: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor,
                                                torch.Tensor, torch.Tensor]:
    """-> (grad_input, grad_offset, grad_mask, grad_weight, grad_bias); grad_mask / grad_bias are
    0-element tensors when the op has no mask / bias (the reference's "fake tensor" convention,
    modulated_deform_conv.py:19-21)."""
    nd, _ = _validate(input, offset, mask, weight, bias, stride, padding, dilation, groups,
                      deformable_groups, in_step, grad_output)
    grad_output = grad_output.contiguous()
    input, offset, weight = _input_layout(input), offset.contiguous(), weight.contiguous()
    mask = None if mask is None else mask.contiguous()
    b = input.new_empty(0) if bias is None else bias.contiguous()
    geo = _geometry(weight, stride, padding, dilation, groups, deformable_groups, in_step,
                    bias is not None)
    fn = _entry(nd, mask is not None, True)
    if mask is not None and nd == 2:
        # (the export itself returns grad_weight / grad_bias as two views of one buffer; an operator's returns may not alias)
        return MDCONV_CUDA._modulated2d_backward(False, input, weight, b, offset, mask, grad_output, *geo)
    gi, goff = torch.empty_like(input, memory_format=torch.contiguous_format), torch.empty_like(offset)
    gw, gb = torch.empty_like(weight), torch.empty_like(b)
    with _capi.overwrite_grads():   # fresh buffers: written, not added to
        if mask is not None:
            gm = torch.empty_like(mask)
            fn(input, weight, b, offset, mask, gi, gw, gb, goff, gm, grad_output, *geo)
        else:
            gm = input.new_empty(0)
            fn(input, weight, b, offset, gi, gw, gb, goff, grad_output, *geo)
    return gi, goff, gm, gw, gb


@deform_conv_backward.register_fake
def _(grad_output, input, offset, mask, weight, bias, stride, padding, dilation, groups,
      deformable_groups, in_step):
    _validate(input, offset, mask, weight, bias, stride, padding, dilation, groups,
              deformable_groups, in_step, grad_output)
    gm = input.new_empty(0) if mask is None else torch.empty_like(mask)
    gb = input.new_empty(0) if bias is None else torch.empty_like(bias)
    return (torch.empty_like(input, memory_format=torch.contiguous_format), torch.empty_like(offset), gm,
            torch.empty_like(weight), gb)


def _setup_context(ctx, inputs, output):
    (input, offset, mask, weight, bias, stride, padding, dilation, groups, deformable_groups,
     in_step) = inputs
    ctx.save_for_backward(input, offset, mask, weight, bias)
    ctx.conf = (list(stride), list(padding), list(dilation), groups, deformable_groups, in_step)


def _autograd(ctx, grad_output):
    input, offset, mask, weight, bias = ctx.saved_tensors
    gi, goff, gm, gw, gb = deform_conv_backward(grad_output, input, offset, mask, weight, bias,
                                                *ctx.conf)
    return (gi, goff, gm if mask is not None else None, gw, gb if bias is not None else None,
            None, None, None, None, None, None)


deform_conv.register_autograd(_autograd, setup_context=_setup_context)
