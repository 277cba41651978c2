"""``MDCONV_CUDA`` -- the reference's extension-module surface on top of libmdconv_hip.so.

The reference builds a pybind11 module of this name (setup.py:37) and its Python wrapper calls
eight functions of it positionally (modulated_deform_conv.py:28, 57, 112, 142, 194, 225, 281,
313; only two are actually registered in the reference snapshot, mdeformable_conv.cu:460-465 --
SURVEY.md R2).  This module exports all eight with the same positional signatures, argument
meaning, return values and error behaviour (RuntimeError for non-contiguous tensors and for
kernel/channel mismatches, mdeformable_conv.cu:127-148), and forwards to the C ABI
(include/mdconv.h) through ctypes.  Torch is plumbing here: device memory, the current stream,
the caching allocator for the scratch workspace.

Put this directory on ``sys.path`` (or ``import modulated_deform_conv_amd.MDCONV_CUDA as
MDCONV_CUDA``) and the reference's own ``modulated_deform_conv.py`` runs unchanged.
"""
import ctypes

import torch

from . import _capi
from .distributed import fused_grad_buffers

_DTYPES = {torch.float32: _capi.F32, torch.float16: _capi.F16, torch.float64: _capi.F64,
           torch.bfloat16: _capi.BF16}


def _is_channels_last(t):
    """`t` is a dense channels-last tensor the native 16-bit kernels can gather from directly
    (torch.channels_last / channels_last_3d, fp16 / bf16, C a multiple of 32)."""
    if t.dim() not in (4, 5) or t.dtype not in (torch.float16, torch.bfloat16) or t.shape[1] % 32:
        return False
    fmt = torch.channels_last if t.dim() == 4 else torch.channels_last_3d
    return t.is_contiguous(memory_format=fmt)


def _check_contig(**tensors):
    for name, t in tensors.items():
        # extension of the reference's check (mdeformable_conv.cu:127-131): a channels-last `input`
        # is accepted where the kernels consume that layout anyway
        if not t.is_contiguous() and not (name == "input" and _is_channels_last(t)):
            raise RuntimeError("%s tensor has to be contiguous" % name)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr() if t.numel() > 0 else 0)


def _desc(nd, modulated, input, weight, ksz, stride, pad, dil, group, deformable_group, in_step,
          with_bias):
    if input.dim() != nd + 2 or weight.dim() != nd + 2:
        raise RuntimeError("expected %d-D input and weight, got %d-D and %d-D"
                           % (nd + 2, input.dim(), weight.dim()))
    if not input.is_cuda:
        raise NotImplementedError  # reference: modulated_deform_conv.py:22-23
    if input.dtype not in _DTYPES:
        raise RuntimeError('"deform_conv" not implemented for %s' % input.dtype)
    if tuple(weight.shape[2:]) != tuple(ksz):
        raise RuntimeError("Input shape and kernel shape wont match: (%s vs %s)."
                           % ("x".join(map(str, ksz)), "x".join(map(str, weight.shape[2:]))))
    if input.shape[1] != weight.shape[1] * group:
        raise RuntimeError("Input shape and kernel channels wont match: (%d vs %d)."
                           % (input.shape[1], weight.shape[1] * group))
    d = _capi.MdconvDesc()
    d.ndim, d.modulated, d.dtype = nd | _capi.DESC_V2, int(modulated), _DTYPES[input.dtype]
    d.accumulate, d.input_layout, d.path = _capi.accumulate_mode(), 0, _capi.PATH_AUTO
    d.batch, d.c_in, d.c_out = input.shape[0], input.shape[1], weight.shape[0]
    fill = lambda v, f: tuple(int(x) for x in v) + (f,) * (3 - nd)
    d.in_sz = (ctypes.c_int * 3)(*fill(input.shape[2:], 1))
    d.k_sz = (ctypes.c_int * 3)(*fill(ksz, 1))
    d.stride = (ctypes.c_int * 3)(*fill(stride, 1))
    d.pad = (ctypes.c_int * 3)(*fill(pad, 0))
    d.dil = (ctypes.c_int * 3)(*fill(dil, 1))
    d.groups, d.dgroups, d.in_step, d.with_bias = int(group), int(deformable_group), int(in_step), int(bool(with_bias))
    return d


def _layout(d, input, backward):
    """A channels-last `input` stays as it is where the kernels of this direction gather from that
    layout (include/mdconv.h: mdconv_input_layout_supported -- the native 16-bit backward covers
    fewer shapes than the forward, and MDCONV_PATH / MDCONV_HP can switch those kernels off);
    otherwise the call runs on a contiguous copy, like any other PyTorch operator would."""
    if input.is_contiguous() or _capi.lib().mdconv_input_layout_supported(ctypes.byref(d), 1, int(backward)):
        return input
    return input.contiguous()


def _out_shape(d, nd):
    L = _capi.lib()
    return tuple(L.mdconv_out_size(ctypes.byref(d), a) for a in range(nd))


def _check_side(d, nd, K, offset, mask, other, other_name, osz):
    """Shape/dtype/device checks the reference omits (a wrong shape would read out of bounds)."""
    exp_off = (d.batch, d.dgroups * nd * K) + osz
    if tuple(offset.shape) != exp_off:
        raise RuntimeError("offset shape %s, expected %s" % (tuple(offset.shape), exp_off))
    if mask is not None:
        exp_m = (d.batch, d.dgroups * K) + osz
        if tuple(mask.shape) != exp_m:
            raise RuntimeError("mask shape %s, expected %s" % (tuple(mask.shape), exp_m))
    n_out = d.batch * d.c_out * _prod(osz)
    if other is not None and other.numel() != n_out:   # the reference .view()s it to this shape
        raise RuntimeError("%s has %d elements, expected %s" % (other_name, other.numel(),
                                                                (d.batch, d.c_out) + osz))


def _same(ref, **tensors):
    for name, t in tensors.items():
        if t is None or t.numel() == 0:
            continue
        if t.dtype != ref.dtype or t.device != ref.device:
            raise RuntimeError("%s must have the dtype/device of input (%s/%s), got %s/%s"
                               % (name, ref.dtype, ref.device, t.dtype, t.device))


def _run(fn_name, d, backward, args_before_ws, input):
    L = _capi.lib()
    d.input_layout = int(not input.is_contiguous() and _is_channels_last(input))
    with torch.cuda.device(input.device):
        ws_bytes = L.mdconv_workspace_bytes(ctypes.byref(d), int(backward))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=input.device) if ws_bytes else None
        stream = torch.cuda.current_stream().cuda_stream
        rc = getattr(L, fn_name)(ctypes.byref(d), *args_before_ws,
                                 ctypes.c_void_p(ws.data_ptr() if ws is not None else 0),
                                 ctypes.c_size_t(ws_bytes), ctypes.c_void_p(stream))
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (fn_name, rc, _capi.last_error()))


def _prod(v):
    p = 1
    for x in v:
        p *= int(x)
    return p


def _forward(nd, modulated, fn_name, input, weight, bias, offset, mask, output, ksz, stride, pad,
             dil, group, deformable_group, in_step, with_bias):
    tensors = dict(input=input, weight=weight, bias=bias, offset=offset)
    if modulated:
        tensors["mask"] = mask
    if output is not None:
        tensors["output"] = output
    _check_contig(**tensors)
    d = _desc(nd, modulated, input, weight, ksz, stride, pad, dil, group, deformable_group, in_step,
              with_bias)
    input = _layout(d, input, False)
    osz = _out_shape(d, nd)
    _check_side(d, nd, _prod(ksz), offset, mask if modulated else None, output, "output", osz)
    _same(input, weight=weight, offset=offset, mask=mask if modulated else None,
          bias=bias if with_bias else None, output=output)
    if with_bias and bias.numel() != d.c_out:
        raise RuntimeError("bias has %d elements, expected %d" % (bias.numel(), d.c_out))
    if output is None:
        output = torch.empty((d.batch, d.c_out) + osz, dtype=input.dtype, device=input.device)
    args = [_ptr(input), _ptr(weight), _ptr(bias), _ptr(offset)]
    if modulated:
        args.append(_ptr(mask))
    args.append(_ptr(output))
    _run(fn_name, d, False, args, input)
    return output


# --------------------------------------------------------------------------------- 2-D, DCNv1
def deform_conv2d_forward_cuda(input, weight, bias, offset, output, kernel_h, kernel_w, stride_h,
                               stride_w, pad_h, pad_w, dilation_h, dilation_w, group,
                               deformable_group, in_step, with_bias):
    """reference deformable_conv.cu:117-123; writes ``output`` in place, returns 0."""
    _forward(2, False, "mdconv_deform_conv2d_forward", input, weight, bias, offset, None, output,
             (kernel_h, kernel_w), (stride_h, stride_w), (pad_h, pad_w), (dilation_h, dilation_w),
             group, deformable_group, in_step, with_bias)
    return 0


def deform_conv2d_backward_cuda(input, weight, bias, offset, grad_input, grad_weight, grad_bias,
                                grad_offset, grad_output, kernel_h, kernel_w, stride_h, stride_w,
                                pad_h, pad_w, dilation_h, dilation_w, group, deformable_group,
                                in_step, with_bias):
    """reference deformable_conv.cu:327-333; accumulates into the four grad tensors, returns 0."""
    _check_contig(input=input, weight=weight, bias=bias, offset=offset, grad_input=grad_input,
                  grad_weight=grad_weight, grad_bias=grad_bias, grad_offset=grad_offset,
                  grad_output=grad_output)
    d = _desc(2, False, input, weight, (kernel_h, kernel_w), (stride_h, stride_w), (pad_h, pad_w),
              (dilation_h, dilation_w), group, deformable_group, in_step, with_bias)
    input = _layout(d, input, True)
    osz = _out_shape(d, 2)
    _check_side(d, 2, kernel_h * kernel_w, offset, None, grad_output, "grad_output", osz)
    _backward_checks(input, weight, offset, None, grad_input, grad_weight, grad_bias, grad_offset,
                     None, grad_output, d, with_bias)
    _run("mdconv_deform_conv2d_backward", d, True,
         [_ptr(input), _ptr(weight), _ptr(bias), _ptr(offset), _ptr(grad_input), _ptr(grad_weight),
          _ptr(grad_bias), _ptr(grad_offset), _ptr(grad_output)], input)
    return 0


def _backward_checks(input, weight, offset, mask, grad_input, grad_weight, grad_bias, grad_offset,
                     grad_mask, grad_output, d, with_bias):
    _same(input, weight=weight, offset=offset, mask=mask, grad_input=grad_input,
          grad_weight=grad_weight, grad_offset=grad_offset, grad_mask=grad_mask,
          grad_output=grad_output, grad_bias=grad_bias if with_bias else None)
    for name, g, ref in (("grad_input", grad_input, input), ("grad_weight", grad_weight, weight),
                         ("grad_offset", grad_offset, offset), ("grad_mask", grad_mask, mask)):
        if ref is not None and g.numel() != ref.numel():
            raise RuntimeError("%s has %d elements, expected %d" % (name, g.numel(), ref.numel()))
    if with_bias and grad_bias.numel() != d.c_out:
        raise RuntimeError("grad_bias has %d elements, expected %d" % (grad_bias.numel(), d.c_out))


# --------------------------------------------------------------------------------- 2-D, DCNv2
def modulated_deform_conv2d_forward_cuda(input, weight, bias, offset, mask, kernel_h, kernel_w,
                                         stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
                                         group, deformable_group, in_step, with_bias):
    """reference mdeformable_conv.cu:120-126; returns a NEW tensor [B, O, Ho, Wo]."""
    return _forward(2, True, "mdconv_modulated_deform_conv2d_forward", input, weight, bias, offset,
                    mask, None, (kernel_h, kernel_w), (stride_h, stride_w), (pad_h, pad_w),
                    (dilation_h, dilation_w), group, deformable_group, in_step, with_bias)


def modulated_deform_conv2d_backward_cuda(input, weight, bias, offset, mask, grad_output, kernel_h,
                                          kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h,
                                          dilation_w, group, deformable_group, in_step, with_bias):
    """reference mdeformable_conv.cu:361-366, 456; returns the tuple
    (grad_input, grad_offset, grad_mask, grad_weight, grad_bias) of new tensors."""
    return _modulated2d_backward(True, input, weight, bias, offset, mask, grad_output, kernel_h, kernel_w, stride_h,
                                 stride_w, pad_h, pad_w, dilation_h, dilation_w, group, deformable_group, in_step, with_bias)


def _modulated2d_backward(fused, input, weight, bias, offset, mask, grad_output, kernel_h, kernel_w, stride_h, stride_w,
                          pad_h, pad_w, dilation_h, dilation_w, group, deformable_group, in_step, with_bias):
    """`fused`: grad_weight and grad_bias are views of ONE buffer (the data-parallel exchange reduces it in place);
    False for torch.library operators, whose returns must not share storage (ops.py)."""
    _check_contig(input=input, weight=weight, bias=bias, offset=offset, mask=mask)
    grad_output = grad_output.contiguous()
    d = _desc(2, True, input, weight, (kernel_h, kernel_w), (stride_h, stride_w), (pad_h, pad_w),
              (dilation_h, dilation_w), group, deformable_group, in_step, with_bias)
    input = _layout(d, input, True)
    osz = _out_shape(d, 2)
    _check_side(d, 2, kernel_h * kernel_w, offset, mask, grad_output, "grad_output", osz)
    # the reference allocates zeros here (mdeformable_conv.cu:404-411) and adds into them; this
    # entry point owns its results, so it allocates uninitialised memory and asks the library to
    # WRITE the gradients (mdconv_desc.accumulate = 0: no zero fills, no read-modify-write)
    grad_input = torch.empty_like(input, memory_format=torch.contiguous_format)
    grad_offset = torch.empty_like(offset)
    grad_mask = torch.empty_like(mask)
    # grad_weight || grad_bias live in ONE flat buffer: the data-parallel exchange is then a single in-place all-reduce
    # (distributed.py: fused_grad_buffers / FusedGradAllReduce)
    grad_weight, grad_bias = fused_grad_buffers(weight, bias) if fused else (torch.empty_like(weight), torch.empty_like(bias))
    _backward_checks(input, weight, offset, mask, grad_input, grad_weight, grad_bias, grad_offset,
                     grad_mask, grad_output, d, with_bias)
    d.accumulate = 0
    _run("mdconv_modulated_deform_conv2d_backward", d, True,
         [_ptr(input), _ptr(weight), _ptr(bias), _ptr(offset), _ptr(mask), _ptr(grad_output),
          _ptr(grad_input), _ptr(grad_offset), _ptr(grad_mask), _ptr(grad_weight),
          _ptr(grad_bias)], input)
    return (grad_input, grad_offset, grad_mask, grad_weight, grad_bias)


# --------------------------------------------------------------------------------- 3-D, DCNv1
def deform_conv3d_forward_cuda(input, weight, bias, offset, output, kernel_h, kernel_w, kernel_l,
                               stride_h, stride_w, stride_l, pad_h, pad_w, pad_l, dilation_h,
                               dilation_w, dilation_l, group, deformable_group, in_step, with_bias):
    """reference deformable_conv3d.cu:160-167; writes ``output`` in place, returns 0."""
    _forward(3, False, "mdconv_deform_conv3d_forward", input, weight, bias, offset, None, output,
             (kernel_h, kernel_w, kernel_l), (stride_h, stride_w, stride_l), (pad_h, pad_w, pad_l),
             (dilation_h, dilation_w, dilation_l), group, deformable_group, in_step, with_bias)
    return 0


def deform_conv3d_backward_cuda(input, weight, bias, offset, grad_input, grad_weight, grad_bias,
                                grad_offset, grad_output, kernel_h, kernel_w, kernel_l, stride_h,
                                stride_w, stride_l, pad_h, pad_w, pad_l, dilation_h, dilation_w,
                                dilation_l, group, deformable_group, in_step, with_bias):
    """reference deformable_conv3d.cu:434-442; accumulates, returns 0."""
    _check_contig(input=input, weight=weight, bias=bias, offset=offset, grad_input=grad_input,
                  grad_weight=grad_weight, grad_bias=grad_bias, grad_offset=grad_offset,
                  grad_output=grad_output)
    ksz = (kernel_h, kernel_w, kernel_l)
    d = _desc(3, False, input, weight, ksz, (stride_h, stride_w, stride_l), (pad_h, pad_w, pad_l),
              (dilation_h, dilation_w, dilation_l), group, deformable_group, in_step, with_bias)
    input = _layout(d, input, True)
    osz = _out_shape(d, 3)
    _check_side(d, 3, _prod(ksz), offset, None, grad_output, "grad_output", osz)
    _backward_checks(input, weight, offset, None, grad_input, grad_weight, grad_bias, grad_offset,
                     None, grad_output, d, with_bias)
    _run("mdconv_deform_conv3d_backward", d, True,
         [_ptr(input), _ptr(weight), _ptr(bias), _ptr(offset), _ptr(grad_input), _ptr(grad_weight),
          _ptr(grad_bias), _ptr(grad_offset), _ptr(grad_output)], input)
    return 0


# --------------------------------------------------------------------------------- 3-D, DCNv2
def modulated_deform_conv3d_forward_cuda(input, weight, bias, offset, mask, output, kernel_h,
                                         kernel_w, kernel_l, stride_h, stride_w, stride_l, pad_h,
                                         pad_w, pad_l, dilation_h, dilation_w, dilation_l, group,
                                         deformable_group, in_step, with_bias):
    """reference mdeformable_conv3d.cu:170-177; writes ``output`` in place, returns 0."""
    _forward(3, True, "mdconv_modulated_deform_conv3d_forward", input, weight, bias, offset, mask,
             output, (kernel_h, kernel_w, kernel_l), (stride_h, stride_w, stride_l),
             (pad_h, pad_w, pad_l), (dilation_h, dilation_w, dilation_l), group, deformable_group,
             in_step, with_bias)
    return 0


def modulated_deform_conv3d_backward_cuda(input, weight, bias, offset, mask, grad_input,
                                          grad_weight, grad_bias, grad_offset, grad_mask,
                                          grad_output, kernel_h, kernel_w, kernel_l, stride_h,
                                          stride_w, stride_l, pad_h, pad_w, pad_l, dilation_h,
                                          dilation_w, dilation_l, group, deformable_group, in_step,
                                          with_bias):
    """reference mdeformable_conv3d.cu:443-451; accumulates, returns 0."""
    _check_contig(input=input, weight=weight, bias=bias, offset=offset, mask=mask,
                  grad_input=grad_input, grad_weight=grad_weight, grad_bias=grad_bias,
                  grad_offset=grad_offset, grad_mask=grad_mask, grad_output=grad_output)
    ksz = (kernel_h, kernel_w, kernel_l)
    d = _desc(3, True, input, weight, ksz, (stride_h, stride_w, stride_l), (pad_h, pad_w, pad_l),
              (dilation_h, dilation_w, dilation_l), group, deformable_group, in_step, with_bias)
    input = _layout(d, input, True)
    osz = _out_shape(d, 3)
    _check_side(d, 3, _prod(ksz), offset, mask, grad_output, "grad_output", osz)
    _backward_checks(input, weight, offset, mask, grad_input, grad_weight, grad_bias, grad_offset,
                     grad_mask, grad_output, d, with_bias)
    _run("mdconv_modulated_deform_conv3d_backward", d, True,
         [_ptr(input), _ptr(weight), _ptr(bias), _ptr(offset), _ptr(mask), _ptr(grad_input),
          _ptr(grad_weight), _ptr(grad_bias), _ptr(grad_offset), _ptr(grad_mask),
          _ptr(grad_output)], input)
    return 0
