"""The channels-last variants of the fp32 kernels -- forward, GEMM-2 and the line-wide GEMM-1 drain
(with its 3-tap flush groups) -- are selected by shape (3-D always, 2-D from ~8 k output pixels), so
the small 2-D parity cases never reach them on their own.  The selection knobs are read once per
process, hence a child process: the MFMA-eligible parity cases run again with all three forced on
and are compared with the oracle as usual (tests/test_gpu_parity.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_parity_with_channels_last_kernels_forced():
    env = dict(os.environ, MDCONV_FWD_CL="1", MDCONV_BWD_CL="1", MDCONV_BD_CL="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-m", "gpu", "-q", "-x",
                        "-k", "auto_path or mfma_path or overwrite"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
