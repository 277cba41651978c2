"""ctypes loader for libmdconv_hip.so -- the C ABI declared in include/mdconv.h.

There is no CPU fallback: if the HIP library is missing or fails to load, importing the product
raises, loudly.
"""
import ctypes
import os
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MDCONV_LIB") or os.path.join(HERE, "libmdconv_hip.so")

F32, F16, F64, BF16 = 0, 1, 2, 3
PATH_AUTO, PATH_DIRECT, PATH_MFMA = 0, 1, 2
ABI_VERSION = 2
DESC_V2 = 0x100   # MDCONV_DESC_V2: the descriptor carries accumulate / input_layout / path

EXPORTS = (
    "mdconv_abi_version", "mdconv_last_error", "mdconv_out_size", "mdconv_workspace_bytes",
    "mdconv_set_path", "mdconv_last_path", "mdconv_last_kernels",
    "mdconv_profile_enable", "mdconv_profile_read", "mdconv_profile_reset", "mdconv_profile_name",
    "mdconv_stream_wait_weight_ready", "mdconv_stream_wait_weight_ready_on", "mdconv_set_accumulate", "mdconv_set_input_layout",
    "mdconv_input_layout_supported",
    "mdconv_deform_conv2d_forward", "mdconv_deform_conv2d_backward",
    "mdconv_modulated_deform_conv2d_forward", "mdconv_modulated_deform_conv2d_backward",
    "mdconv_deform_conv3d_forward", "mdconv_deform_conv3d_backward",
    "mdconv_modulated_deform_conv3d_forward", "mdconv_modulated_deform_conv3d_backward",
)


class MdconvDesc(ctypes.Structure):
    """Mirror of ``struct mdconv_desc`` (include/mdconv.h, ABI v2: the call modes travel in the descriptor)."""
    _fields_ = [("ndim", ctypes.c_int), ("modulated", ctypes.c_int), ("dtype", ctypes.c_int),
                ("batch", ctypes.c_int), ("c_in", ctypes.c_int), ("c_out", ctypes.c_int),
                ("in_sz", ctypes.c_int * 3), ("k_sz", ctypes.c_int * 3),
                ("stride", ctypes.c_int * 3), ("pad", ctypes.c_int * 3), ("dil", ctypes.c_int * 3),
                ("groups", ctypes.c_int), ("dgroups", ctypes.c_int), ("in_step", ctypes.c_int),
                ("with_bias", ctypes.c_int),
                ("accumulate", ctypes.c_int), ("input_layout", ctypes.c_int), ("path", ctypes.c_int),
                ("reserved", ctypes.c_int * 5)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libmdconv_hip.so is not built (%s). Run `python -m modulated_deform_conv_amd._build` "
                "(needs hipcc); there is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.mdconv_abi_version.restype = ctypes.c_int
        L.mdconv_last_error.restype = ctypes.c_char_p
        L.mdconv_out_size.restype = ctypes.c_int
        L.mdconv_workspace_bytes.restype = ctypes.c_size_t
        L.mdconv_workspace_bytes.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.mdconv_set_path.restype = ctypes.c_int
        L.mdconv_set_path.argtypes = [ctypes.c_int]
        L.mdconv_last_path.restype = ctypes.c_int
        L.mdconv_profile_enable.restype = ctypes.c_int
        L.mdconv_profile_read.restype = ctypes.c_int
        L.mdconv_profile_reset.restype = None
        L.mdconv_profile_name.restype = ctypes.c_char_p
        L.mdconv_profile_name.argtypes = [ctypes.c_int]
        L.mdconv_set_accumulate.restype = ctypes.c_int
        L.mdconv_set_accumulate.argtypes = [ctypes.c_int]
        L.mdconv_set_input_layout.restype = ctypes.c_int
        L.mdconv_set_input_layout.argtypes = [ctypes.c_int]
        L.mdconv_stream_wait_weight_ready.restype = ctypes.c_int
        L.mdconv_stream_wait_weight_ready.argtypes = [ctypes.c_void_p]
        L.mdconv_stream_wait_weight_ready_on.restype = ctypes.c_int
        L.mdconv_stream_wait_weight_ready_on.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.mdconv_last_kernels.restype = ctypes.c_int
        L.mdconv_input_layout_supported.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        for name in EXPORTS[11:]:
            getattr(L, name).restype = ctypes.c_int
        if L.mdconv_abi_version() != ABI_VERSION:
            raise ImportError("libmdconv_hip.so ABI version mismatch")
        _lib = L
    return _lib


def last_error():
    return lib().mdconv_last_error().decode("utf-8", "replace")


def set_path(path):
    """Force the kernel path: 'auto' | 'direct' | 'mfma'.  Returns the previous setting."""
    names = {"auto": PATH_AUTO, "direct": PATH_DIRECT, "mfma": PATH_MFMA}
    prev = lib().mdconv_set_path(names[path] if isinstance(path, str) else int(path))
    return {v: k for k, v in names.items()}[prev]


def last_path():
    return {0: "none", PATH_DIRECT: "direct", PATH_MFMA: "mfma"}[lib().mdconv_last_path()]


def last_kernels():
    """Kernel family of the last call: 'direct' | 'f32' (fp32 MFMA kernels) | 'hp' (native 16-bit)."""
    return {0: "none", 1: "direct", 2: "f32", 3: "hp"}[lib().mdconv_last_kernels()]


PROFILE_SLOTS = 5   # forward GEMM, backward data GEMM, backward weight GEMM, grad_input gather


_modes = threading.local()   # Python-side default of mdconv_desc.accumulate for descriptors built in this thread


def accumulate_mode():
    """Value for ``mdconv_desc.accumulate`` of a descriptor built now (1 unless inside ``overwrite_grads``)."""
    return getattr(_modes, "accumulate", 1)


class overwrite_grads:
    """Context manager: backward entry points of MDCONV_CUDA called inside WRITE their gradients instead of
    adding to them (``mdconv_desc.accumulate = 0``, include/mdconv.h), so the buffers may be torch.empty.
    The mode travels in each call's descriptor; no library state is touched."""

    def __enter__(self):
        self._prev = accumulate_mode()
        _modes.accumulate = 0

    def __exit__(self, *exc):
        _modes.accumulate = self._prev
        return False


def stream_wait_weight_ready(stream, producer=None):
    """Make `stream` (a torch.cuda.Stream) wait until grad_weight / grad_bias of the last backward
    issued on `producer` (a torch.cuda.Stream; default: the most recent backward on the current
    device) are final -- the grad_input gather may still be running.  Works whichever host thread
    issued that backward (autograd worker threads included)."""
    if producer is None:
        rc = lib().mdconv_stream_wait_weight_ready(ctypes.c_void_p(stream.cuda_stream))
    else:
        rc = lib().mdconv_stream_wait_weight_ready_on(ctypes.c_void_p(stream.cuda_stream),
                                                      ctypes.c_void_p(producer.cuda_stream))
    if rc != 0:
        raise RuntimeError(last_error())


def profile_enable(on=True):
    return bool(lib().mdconv_profile_enable(int(on)))


def profile_reset():
    lib().mdconv_profile_reset()


def profile_read():
    """{kernel name: (launches, average ms)} -- call after torch.cuda.synchronize()."""
    out = {}
    for which in range(PROFILE_SLOTS):
        tot = ctypes.c_double(0)
        n = lib().mdconv_profile_read(which, ctypes.byref(tot))
        name = lib().mdconv_profile_name(which).decode()
        if n and name:
            out[name] = (n, tot.value / n)
    return out
