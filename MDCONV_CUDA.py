"""Top-level shim with the reference's extension-module name (reference setup.py:37).

``import MDCONV_CUDA`` -- which is what the reference's modulated_deform_conv.py does at line 7 --
resolves to the ctypes binding of libmdconv_hip.so."""
from modulated_deform_conv_amd.MDCONV_CUDA import (  # noqa: F401
    deform_conv2d_backward_cuda, deform_conv2d_forward_cuda, deform_conv3d_backward_cuda,
    deform_conv3d_forward_cuda, modulated_deform_conv2d_backward_cuda,
    modulated_deform_conv2d_forward_cuda, modulated_deform_conv3d_backward_cuda,
    modulated_deform_conv3d_forward_cuda)
