"""Packaging proof (reference setup.py:36-41): `pip install .` builds the
HIP library and installs the package plus the two top-level modules with the reference's import
names -- MDCONV_CUDA (the extension module, reference setup.py:37) and modulated_deform_conv (the
Python wrapper, setup.py:41) -- and user code written against the reference imports them unchanged.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_PROBE = r"""
import sys
sys.path.insert(0, %r)
import MDCONV_CUDA, modulated_deform_conv as mdc
import modulated_deform_conv_amd, os
assert os.path.dirname(modulated_deform_conv_amd.__file__).startswith(%r), modulated_deform_conv_amd.__file__
names = ["deform_conv2d_forward_cuda", "deform_conv2d_backward_cuda", "modulated_deform_conv2d_forward_cuda",
         "modulated_deform_conv2d_backward_cuda", "deform_conv3d_forward_cuda", "deform_conv3d_backward_cuda",
         "modulated_deform_conv3d_forward_cuda", "modulated_deform_conv3d_backward_cuda"]
assert all(callable(getattr(MDCONV_CUDA, n)) for n in names)
for n in ("DeformConv2d", "ModulatedDeformConv2d", "DeformConv3d", "ModulatedDeformConv3d",
          "DeformConv2dPack", "ModulatedDeformConv2dPack", "deform_conv2d", "modulated_deform_conv2d"):
    assert hasattr(mdc, n), n
from modulated_deform_conv_amd import _capi
assert os.path.exists(_capi.LIB_PATH) and _capi.lib().mdconv_abi_version() == 2
%s
print("PACKAGING_OK")
"""

# my_test.py's scenario (reference my_test.py:5-24) through the installed top-level names
_GPU_SCENARIO = r"""
import torch
x = torch.ones(1, 1, 5, 5).cuda(); off = torch.zeros(1, 18, 5, 5).cuda(); m = torch.ones(1, 9, 5, 5).cuda()
w = torch.ones(1, 1, 3, 3).cuda().requires_grad_(); b = torch.zeros(1).cuda()
for out in (mdc.deform_conv2d(x, off, w, b, 1, 1), mdc.modulated_deform_conv2d(x, off, m, w, b, 1, 1)):
    assert out.sum().item() == 169, out
    out.sum().backward()
    assert w.grad.flatten().tolist() == [16., 20, 16, 20, 25, 20, 16, 20, 16]
    w.grad = None
"""


def _install(tmp_path):
    target = str(tmp_path / "site")
    r = subprocess.run([sys.executable, "-m", "pip", "install", "--no-build-isolation", "--no-deps", "--no-index",
                        "--target", target, ROOT], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return target


def _probe(target, extra):
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    r = subprocess.run([sys.executable, "-c", _PROBE % (target, target, extra)], capture_output=True, text=True,
                       cwd="/", env=env, timeout=600)
    assert "PACKAGING_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_pip_install_and_import_drop_in_names(tmp_path):
    _probe(_install(tmp_path), "")


@pytest.mark.gpu
def test_installed_drop_in_names_run_my_test_scenario(tmp_path):
    _probe(_install(tmp_path), _GPU_SCENARIO)
