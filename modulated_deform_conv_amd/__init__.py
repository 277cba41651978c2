"""MI355X-native (gfx950) deformable convolution: the hot path of CHONSPQX/modulated-deform-conv
behind the reference's own operator surface.

    modulated_deform_conv_amd.MDCONV_CUDA              the 8 extension-module entry points
    modulated_deform_conv_amd.modulated_deform_conv    autograd Functions + nn.Modules
    modulated_deform_conv_amd.ops                      torch.library ops mdconv::deform_conv[_backward]
                                                       (fake kernels: torch.compile / export)
    modulated_deform_conv_amd.distributed              batch-sharded multi-GPU helper (RCCL)

All compute runs in libmdconv_hip.so (hand-written HIP, C ABI in include/mdconv.h); there is no
CPU or PyTorch fallback.
"""
from . import _capi  # noqa: F401  (does not load the library until first use)

__version__ = "0.1.0"
