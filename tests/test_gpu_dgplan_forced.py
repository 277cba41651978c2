"""Deformable groups the fp32 matrix kernels do not tile have two plans (mfma_kernels.hip): ONE problem with every group padded
to a tileable size (the default since round 6, one conv group) or DG single-group slices (conv groups, and every such shape
until round 6).  The plan is chosen once per process (MDCONV_DG_PLAN), hence child processes: the split / padded parity cases
run under both and are compared with the oracle as usual (tests/test_gpu_parity.py), so the slices stay covered for the
one-conv-group shapes that no longer reach them by default."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("plan", ["split", "pad"])
def test_parity_with_the_deformable_group_plan_forced(plan):
    env = dict(os.environ, MDCONV_DG_PLAN=plan)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-m", "gpu", "-q", "-x",
                        "-k", "mfma_split or idle" if plan == "split" else "mfma_split or mfma_pad or idle"],   # (the slices take groups of 16+ channels)
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("on", ["1", "0"])
def test_parity_with_channel_padding_forced(on):
    """One deformable group and C_in not a multiple of 64: padded to 64 channels the shape runs on the channels-last fp32 kernels
    (mfma_kernels.hip, pad_channels_preferred: 3-D from 2048 pixels, narrow 2-D from 8192).  MDCONV_PAD_CHANNELS = 1 takes the
    plan for every eligible shape, 0 for none: the matrix-path parity cases and the fp32 random shapes run under both."""
    env = dict(os.environ, MDCONV_PAD_CHANNELS=on)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "tests/test_gpu_fuzz.py", "-m", "gpu", "-q", "-x",
                        "-k", "(mfma_path or auto_path or overwrite or non_finite or mfma_equals_direct or wide_geometry_fp32)"
                              + ("" if on == "1" else " and not mfma_padt and not mfma_padn and not mfma_padg")],   # (without the plan those shapes run on the shape-generic kernels)
                       cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
