"""Top-level shim with the reference's Python module name (reference setup.py:41): the same
classes and functional aliases, backed by the MI355X library."""
from modulated_deform_conv_amd.modulated_deform_conv import *  # noqa: F401,F403
from modulated_deform_conv_amd.modulated_deform_conv import __all__  # noqa: F401
