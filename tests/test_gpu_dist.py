"""The RCCL leg of the data-parallel plan on ONE GPU (world_size 1): process-group set-up, the
communication stream waiting on the library's weights-ready event, the fused all-reduce and the
re-join -- everything `bench.py --gpus N` does per step except having peers."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29917")
os.environ["NCCL_DEBUG"] = "WARN"
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from modulated_deform_conv_amd import MDCONV_CUDA as M
from modulated_deform_conv_amd.distributed import FusedGradAllReduce
from tests.cases import CASE_BY_NAME, make_inputs
case = CASE_BY_NAME["cfg2s_mdcn2d_c64_28x28_b4"]
t = make_inputs(case, device="cuda")
geo = (3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 64, True)
ref = M.modulated_deform_conv2d_backward_cuda(t["input"], t["weight"], t["bias"], t["offset"], t["mask"],
                                              t["grad_output"], *geo)
torch.cuda.synchronize()
sync = FusedGradAllReduce()
for _ in range(3):
    g = M.modulated_deform_conv2d_backward_cuda(t["input"], t["weight"], t["bias"], t["offset"], t["mask"],
                                                t["grad_output"], *geo)
    sync.reduce_overlapped(g[3], g[4])
torch.cuda.synchronize()
assert torch.equal(g[3], ref[3]) and torch.equal(g[4], ref[4]), "all-reduce over one rank must be the identity"
dist.destroy_process_group()
print("DIST-OK")
'''


def test_overlapped_allreduce_on_rccl_world_size_1():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", SCRIPT], cwd=root, capture_output=True, text=True, timeout=300)
    assert "DIST-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
