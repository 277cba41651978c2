"""ctypes binding of the CPU oracle (``oracle/mdconv_oracle.c``).

TEST INFRASTRUCTURE ONLY: importable from ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  The product package never imports this module.

The oracle restates the reference's four ops; parity pin status is in the
header of ``mdconv_oracle.h`` (PARITY UNPINNED against reference-executed vectors).
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmdconv_oracle.so")

DCN2D, MDCN2D, DCN3D, MDCN3D = 0, 1, 2, 3
OP_NAMES = {DCN2D: "deform_conv2d", MDCN2D: "modulated_deform_conv2d",
            DCN3D: "deform_conv3d", MDCN3D: "modulated_deform_conv3d"}


class OracleDesc(ctypes.Structure):
    _fields_ = [("op", ctypes.c_int), ("batch", ctypes.c_int), ("c_in", ctypes.c_int),
                ("c_out", ctypes.c_int), ("in_sz", ctypes.c_int * 3), ("k_sz", ctypes.c_int * 3),
                ("stride", ctypes.c_int * 3), ("pad", ctypes.c_int * 3), ("dil", ctypes.c_int * 3),
                ("groups", ctypes.c_int), ("dgroups", ctypes.c_int), ("in_step", ctypes.c_int),
                ("with_bias", ctypes.c_int)]


def build(force=False):
    """Compile the oracle with gcc (``make -C oracle``)."""
    srcs = [os.path.join(_HERE, f) for f in
            ("mdconv_oracle.c", "mdconv_oracle_body.h", "mdconv_oracle.h", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_num_threads.restype = ctypes.c_int
        _lib.oracle_out_size.restype = ctypes.c_int
        for name in ("oracle_forward_f32", "oracle_forward_f64", "oracle_backward_f32",
                     "oracle_backward_f64"):
            getattr(_lib, name).restype = ctypes.c_int
    return _lib


def num_threads():
    return int(lib().oracle_num_threads())


def _triple(v, nd, fill):
    if isinstance(v, int):
        v = (v,) * nd
    v = tuple(int(x) for x in v)
    assert len(v) == nd
    return v + (fill,) * (3 - nd)


def make_desc(op, input, weight, stride, padding, dilation, groups, dgroups, in_step, with_bias):
    nd = 3 if op in (DCN3D, MDCN3D) else 2
    d = OracleDesc()
    d.op = op
    d.batch, d.c_in = int(input.shape[0]), int(input.shape[1])
    d.c_out = int(weight.shape[0])
    d.in_sz = (ctypes.c_int * 3)(*_triple(tuple(input.shape[2:]), nd, 1))
    d.k_sz = (ctypes.c_int * 3)(*_triple(tuple(weight.shape[2:]), nd, 1))
    d.stride = (ctypes.c_int * 3)(*_triple(stride, nd, 1))
    d.pad = (ctypes.c_int * 3)(*_triple(padding, nd, 0))
    d.dil = (ctypes.c_int * 3)(*_triple(dilation, nd, 1))
    d.groups, d.dgroups, d.in_step, d.with_bias = int(groups), int(dgroups), int(in_step), int(with_bias)
    return d, nd


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _prep(t, dtype):
    return None if t is None else t.detach().to(device="cpu", dtype=dtype).contiguous()


def out_shape(d, nd):
    L = lib()
    return tuple(int(L.oracle_out_size(ctypes.byref(d), a)) for a in range(nd))


def forward(op, input, weight, bias, offset, mask=None, stride=1, padding=0, dilation=1,
            groups=1, dgroups=1, in_step=64, dtype=None):
    """Oracle forward; CPU tensors in, CPU tensor out (computed in ``dtype``: fp32 or fp64)."""
    dtype = dtype or (torch.float64 if input.dtype == torch.float64 else torch.float32)
    with_bias = bias is not None and bias.numel() > 0
    d, nd = make_desc(op, input, weight, stride, padding, dilation, groups, dgroups, in_step, with_bias)
    x, w, o = _prep(input, dtype), _prep(weight, dtype), _prep(offset, dtype)
    b = _prep(bias, dtype) if with_bias else None
    m = _prep(mask, dtype) if op in (MDCN2D, MDCN3D) else None
    out = torch.empty((d.batch, d.c_out) + out_shape(d, nd), dtype=dtype)
    fn = lib().oracle_forward_f64 if dtype == torch.float64 else lib().oracle_forward_f32
    rc = fn(ctypes.byref(d), _ptr(x), _ptr(w), _ptr(b), _ptr(o), _ptr(m), _ptr(out))
    if rc != 0:
        raise RuntimeError("oracle_forward: shape error")
    return out


def backward(op, input, weight, bias, offset, mask, grad_output, stride=1, padding=0, dilation=1,
             groups=1, dgroups=1, in_step=64, dtype=None, intermediates=None):
    """Oracle backward from zero-initialised grads.

    ``intermediates`` = torch.float16 / torch.bfloat16: the reference's `columns` / `grad_columns` buffers, which are
    tensors of the input's type (mdeformable_conv.cu:396-397), are rounded to that type where the reference stores them;
    everything else is computed in ``dtype``.  Returns dict(grad_input, grad_offset, grad_mask|None, grad_weight, grad_bias|None)."""
    dtype = dtype or (torch.float64 if input.dtype == torch.float64 else torch.float32)
    with_bias = bias is not None and bias.numel() > 0
    modulated = op in (MDCN2D, MDCN3D)
    d, nd = make_desc(op, input, weight, stride, padding, dilation, groups, dgroups, in_step, with_bias)
    x, w, o = _prep(input, dtype), _prep(weight, dtype), _prep(offset, dtype)
    m = _prep(mask, dtype) if modulated else None
    go = _prep(grad_output, dtype)
    gx, gw, goff = torch.zeros_like(x), torch.zeros_like(w), torch.zeros_like(o)
    gm = torch.zeros_like(m) if modulated else None
    gb = torch.zeros(d.c_out, dtype=dtype) if with_bias else None
    fn = lib().oracle_backward_f64 if dtype == torch.float64 else lib().oracle_backward_f32
    lib().oracle_set_intermediate_rounding({None: 0, torch.float16: 1, torch.bfloat16: 2}[intermediates])
    try:
        rc = fn(ctypes.byref(d), _ptr(x), _ptr(w), _ptr(o), _ptr(m), _ptr(go), _ptr(gx), _ptr(gw),
                _ptr(gb), _ptr(goff), _ptr(gm))
    finally:
        lib().oracle_set_intermediate_rounding(0)
    if rc != 0:
        raise RuntimeError("oracle_backward: shape error")
    return dict(grad_input=gx, grad_offset=goff, grad_mask=gm, grad_weight=gw, grad_bias=gb)
