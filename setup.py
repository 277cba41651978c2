"""Packaging: builds libmdconv_hip.so with hipcc (gfx950) and installs

  * the package ``modulated_deform_conv_amd`` (C-ABI library + bindings), and
  * two top-level shim modules with the reference's import names, ``MDCONV_CUDA`` and
    ``modulated_deform_conv`` (reference setup.py:37, :41), so existing user code keeps working.

    pip install .        (needs /opt/rocm hipcc; no CUDA, no hipify)
"""
import os
import shutil
import sys

from setuptools import setup
from setuptools.command.build_py import build_py

HERE = os.path.dirname(os.path.abspath(__file__))


class BuildWithHip(build_py):
    def run(self):
        sys.path.insert(0, HERE)
        from modulated_deform_conv_amd import _build
        lib = _build.build()
        super().run()
        dst = os.path.join(self.build_lib, "modulated_deform_conv_amd")
        os.makedirs(dst, exist_ok=True)
        shutil.copy2(lib, dst)


setup(
    name="modulated_deform_conv_amd",
    version="0.1.0",
    description="MI355X-native (gfx950) deformable / modulated deformable convolution 2-D and 3-D",
    packages=["modulated_deform_conv_amd"],
    package_data={"modulated_deform_conv_amd": ["csrc/*.hip", "csrc/*.hpp", "libmdconv_hip.so"]},
    py_modules=["MDCONV_CUDA", "modulated_deform_conv"],
    package_dir={"": "."},
    cmdclass={"build_py": BuildWithHip},
    install_requires=["torch>=2.1"],
)
