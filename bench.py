#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json `metric`).

A step = one forward + one backward of ModulatedDeformConv2d 3x3, C_in = C_out = 256, 56x56,
fp32 (BASELINE.json configs[1]) on synthetic inputs already resident in HBM, through the
MDCONV_CUDA surface (ctypes -> C ABI -> HIP kernels).  With N > 1 GPUs the batch is sharded and
the step ends with the one exchange the path has: the fused [grad_weight || grad_bias]
all-reduce over RCCL.
  --scaling weak   (default) B = 32 per GPU, global batch 32 N
  --scaling strong global B = 32, contiguous shards of 32 / N images per GPU (cfg2
                   at 32 / 16 / 8 / 4 images per GPU on 1 / 2 / 4 / 8 GPUs)

    python bench.py --gpus N --steps K --warmup W [--scaling strong]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (see README / DESIGN.md for the field meanings).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

# cfg2 of BASELINE.json
B, C, O, H, W, KH, KW = 32, 256, 256, 56, 56, 3, 3
K = KH * KW
N_SAMPLES = B * C * K * H * W                      # 231 211 008 "samples" per GPU per step
GEMM_FLOP = 2.0 * O * C * K * B * H * W            # one of the three GEMMs (118.4 GFLOP)
PEAK_F32_MFMA_TFLOPS = 157.3                       # MI355X_MICROARCH.md, dense fp32 matrix peak
# compulsory HBM bytes of one fwd+bwd step
COMPULSORY_BYTES = 553_396_224
HBM_PEAK_GBS = 8000.0


def make_inputs(device, batch=B):
    g = torch.Generator(device="cpu").manual_seed(0)
    rn = lambda *s: torch.randn(*s, generator=g)
    x = rn(batch, C, H, W)
    off = rn(batch, 2 * K, H, W)
    m = torch.sigmoid(rn(batch, K, H, W))
    w = (torch.rand(O, C, KH, KW, generator=g) * 2 - 1) / math.sqrt(C * K)
    b = 0.1 * rn(O)
    go = rn(batch, O, H, W)
    return [t.to(device).contiguous() for t in (x, off, m, w, b, go)]


def kernel_sources_sha16():
    """sha256 (first 16 hex digits) over the kernel sources: identifies the code a PMC profile was
    collected on (the GPU box has no .git, so the commit id is not available there)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "modulated_deform_conv_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.hpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 --pmc summary
    (profiles/rNN*_pmc_summary.json, produced by tools/summarize_profile.py).  Counters cannot be
    collected inside this process; the summary is stamped with the hash of the kernel sources it
    was collected on and is REFUSED (None) when that differs from the sources being timed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not files:
        return None, None
    data = json.load(open(files[-1]))
    stamp = data.get("_kernel_sources_sha16")
    src = "%s (sources %s)" % (os.path.basename(files[-1]), stamp)
    if stamp != kernel_sources_sha16():
        return None, src + " -- stale: kernel sources changed since, traffic withheld"
    for k, v in data.items():
        if k.startswith(kernel):
            return v["hbm_bytes_per_launch"], src
    return None, src


def cpu_baseline(batch=B, iters=3):
    """The oracle (CPU restatement of the reference, kind = "port") on a bounded sample of the
    same workload: cfg2 at its full B = 32, `iters` forward + backward passes, median reported.
    im2col and the three GEMMs use every host thread OpenMP provides; the per-sample gradient
    loop (the reference's atomic scatter, mdeformable_conv.cu:202-318) runs on ONE thread so that
    its accumulation order is the reference's sequential one."""
    import oracle
    oracle.build()
    x, off, m, w, b, go = make_inputs("cpu", batch)
    times = []
    for _ in range(iters):
        t0 = time.perf_counter()
        oracle.forward(oracle.MDCN2D, x, w, b, off, m, 1, 1, 1, 1, 1, 64)
        oracle.backward(oracle.MDCN2D, x, w, b, off, m, go, 1, 1, 1, 1, 1, 64)
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    return {"value": batch * C * K * H * W / dt / 1e9, "unit": "GSamples/s",
            "cores": oracle.num_threads(), "kind": "port",
            "sample": "cfg2 at B=%d, %d x (fwd+bwd), median %.1f s per iteration (all: %s); %d OpenMP "
                      "threads for im2col + GEMMs, 1 thread for the per-sample gradient loop"
                      % (batch, iters, dt, ", ".join("%.1f" % t for t in times), oracle.num_threads())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # under torchrun (RANK set) the collective path is exercised even with one process
    distributed = world > 1 or (os.environ.get("MDCONV_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
            os.environ["NCCL_DEBUG"] = "WARN"   # the image's default prints a banner on stdout
        dist.init_process_group("nccl", device_id=device)

    from modulated_deform_conv_amd import MDCONV_CUDA as M, _capi
    from modulated_deform_conv_amd.distributed import FusedGradAllReduce
    x, off, m, w, b, go = make_inputs(device)
    if args.scaling == "strong":
        # global batch 32: this rank's contiguous shard
        from modulated_deform_conv_amd.distributed import shard_bounds
        lo, hi = shard_bounds(B, world, rank)
        if hi <= lo:
            raise SystemExit("--scaling strong needs world size <= %d" % B)
        x, off, m, go = (t[lo:hi].contiguous() for t in (x, off, m, go))
    local_b = x.shape[0]
    geo = (KH, KW, 1, 1, 1, 1, 1, 1, 1, 1, 64, True)
    reducer = FusedGradAllReduce() if distributed else None

    def step():
        out = M.modulated_deform_conv2d_forward_cuda(x, w, b, off, m, *geo)
        gi, goff, gm, gw, gb = M.modulated_deform_conv2d_backward_cuda(x, w, b, off, m, go, *geo)
        if reducer is not None:
            # RCCL all-reduce of grad_weight || grad_bias on a side stream, released as soon as
            # GEMM-2 / grad_bias are done, i.e. under the grad_input gather of the same backward
            reducer.reduce_overlapped(gw, gb)
        return out, gi

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    paths = _capi.last_path()
    _capi.profile_enable(True)
    _capi.profile_reset()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    _capi.profile_enable(False)
    prof = _capi.profile_read()

    if distributed:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    if rank != 0:
        if distributed:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    global_b = B * world if args.scaling == "weak" else B
    value = global_b * C * K * H * W / (elapsed / args.steps) / 1e9
    flop_scale = local_b / B   # work of one launch on this rank relative to the B = 32 figures
    # dominant kernel = the MFMA GEMM kernel with the largest measured average duration
    dom, (dom_n, dom_ms) = max(prof.items(), key=lambda kv: kv[1][1])
    achieved = GEMM_FLOP * flop_scale / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    traffic, traffic_src = measured_traffic(dom)
    result = {
        "metric": "fwd+bwd GSamples/s, MDCN2d 3x3 C=256 56x56 B=32; %HBM roofline",
        "value": round(value, 3), "unit": "GSamples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "ms_per_step_median": round(step_ms[len(step_ms) // 2], 4),
        "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "ModulatedDeformConv2d 3x3, C_in=C_out=256, 56x56, B=%d per GPU, "
                               "deformable_group=1, fp32, forward+backward (BASELINE.json configs[1])"
                               % local_b,
                   "global_batch": global_b, "parallelism": "dp%d batch-sharded" % world,
                   "kernel_path": paths},
        "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2),
                     "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                     "traffic_source": traffic_src,
                     "flop_per_launch": GEMM_FLOP * flop_scale, "avg_ms": round(dom_ms, 4), "launches": dom_n},
        "kernels_ms": {k: round(v[1], 4) for k, v in prof.items()},
        "hbm_roofline": {"compulsory_bytes": int(COMPULSORY_BYTES * flop_scale),
                         "achieved_GBs": round(COMPULSORY_BYTES * flop_scale / (ms_per_step * 1e-3) / 1e9, 1),
                         "peak_GBs": HBM_PEAK_GBS,
                         "frac": round(COMPULSORY_BYTES * flop_scale / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
    }
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline()
    print(json.dumps(result))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
