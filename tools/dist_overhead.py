#!/usr/bin/env python3
"""Developer aid: cost of the gradient exchange per step at world size 1 (RCCL through torchrun):
no exchange / plain all-reduce on the compute stream / overlapped on the communication stream."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from modulated_deform_conv_amd.distributed import FusedGradAllReduce  # noqa: E402

os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
wl = bench.Workload("cfg2", "cuda")
red = FusedGradAllReduce()


def run(mode, steps=20):
    def step():
        wl.forward()
        gw, gb = wl.backward()
        if mode == "plain":
            red(gw, gb)
        elif mode == "overlapped":
            red.reduce_overlapped(gw, gb)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for mode in ("none", "plain", "overlapped", "none", "plain", "overlapped"):
    print(mode, "%.3f ms" % run(mode))
dist.destroy_process_group()
