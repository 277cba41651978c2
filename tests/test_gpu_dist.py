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
# the binding allocates grad_weight || grad_bias as ONE buffer: the exchange is a single in-place all-reduce
from modulated_deform_conv_amd.distributed import fused_view
assert sync.last_mode == "in-place" and fused_view(g[3], g[4]) is not None and sync._flat is None
# the whole step INCLUDING the exchange captured in a HIP graph: the communication stream joins the capture through the
# library's weights-ready event (bench.py --graph with peers); three replays, results of a plain call
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
def step():
    out = M.modulated_deform_conv2d_forward_cuda(t["input"], t["weight"], t["bias"], t["offset"], t["mask"], *geo)
    gg = M.modulated_deform_conv2d_backward_cuda(t["input"], t["weight"], t["bias"], t["offset"], t["mask"], t["grad_output"], *geo)
    sync.reduce_overlapped(gg[3], gg[4])
    return out, gg
with torch.cuda.stream(side):
    step(); step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, capture_error_mode="thread_local"):
    out_s, g_s = step()
for _ in range(3):
    for x in g_s: x.fill_(float("nan"))
    graph.replay()
torch.cuda.synchronize()
assert torch.equal(g_s[3], ref[3]) and torch.equal(g_s[4], ref[4]) and torch.equal(g_s[1], ref[1]) and torch.equal(g_s[2], ref[2])
assert torch.allclose(g_s[0], ref[0], rtol=1e-5, atol=1e-6)
dist.destroy_process_group()
print("DIST-OK")
'''


def test_overlapped_allreduce_on_rccl_world_size_1():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", SCRIPT], cwd=root, capture_output=True, text=True, timeout=300)
    assert "DIST-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
