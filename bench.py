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
  --graph          the step (forward + backward) is captured once in a HIP graph and replayed; the
                   default for --scaling strong, where a 4-image shard (0.6 ms of kernels) is as
                   much host-bound as kernel-bound when every launch is enqueued from Python

    python bench.py --gpus N --steps K --warmup W [--scaling strong] [--graph]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no RANK in the environment re-executes itself under
torch.distributed.run with N ranks (one per GPU); it exits non-zero when fewer than N GPUs are
visible.  Rank 0 prints ONE JSON line (see README / DESIGN.md for the field meanings); besides the
headline it carries `other_configs`: the per-GPU shards of BASELINE.json configs[2..4] timed in
the same process (N = 1 only).
"""
import argparse
import json
import math
import os
import sys
import time

# multi-process GPU work on this host needs dmabuf IPC (RCCL's hipIpcGetMemHandle fails with the legacy mode); the GPU box
# exports it already -- kept here so that a bare `python bench.py --gpus N` from a clean environment works too
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# peaks: /opt/skills/guides/MI355X_MICROARCH.md (dense; no sparsity figures)
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_F16_MFMA_TFLOPS = 2500.0
HBM_PEAK_GBS = 8000.0

# Per-GPU workloads of BASELINE.json configs[1..4].  `bytes` = compulsory HBM
# bytes of one fwd+bwd step of this shard, `flop` = GEMM FLOP of the step (three GEMMs), `bound` =
# the roofline SURVEY 8d assigns the whole op.
WORKLOADS = {
    "cfg2": dict(nd=2, modulated=True, B=32, C=256, O=256, sp=(56, 56), G=1, DG=1, dil=1, dtype="f32", bias=True,
                 bytes=553_396_224, bound="mfma", peak=PEAK_F32_MFMA_TFLOPS,
                 what="ModulatedDeformConv2d 3x3, C_in=C_out=256, 56x56, deformable_group=1, fp32"),
    "cfg3": dict(nd=2, modulated=True, B=32, C=256, O=256, sp=(56, 56), G=32, DG=4, dil=1, dtype="f16", bias=False,
                 bytes=2_575_545_344 // 8, bound="hbm", peak=HBM_PEAK_GBS,
                 what="ModulatedDeformConv2d 3x3, C=256, 56x56, group=32, deformable_group=4, fp16 (1/8 of B=256)"),
    "cfg4": dict(nd=3, modulated=False, B=8, C=64, O=64, sp=(32, 32, 32), G=1, DG=1, dil=1, dtype="f32", bias=False,
                 bytes=591_675_904, bound="mfma", peak=PEAK_F32_MFMA_TFLOPS,
                 what="DeformConv3d 3x3x3, C=64, 32^3, fp32"),
    "cfg5": dict(nd=3, modulated=True, B=8, C=128, O=128, sp=(16, 64, 64), G=1, DG=1, dil=2, dtype="f16", bias=False,
                 bytes=4_045_963_776 // 4, bound="mfma", peak=PEAK_F16_MFMA_TFLOPS,
                 what="ModulatedDeformConv3d 3x3x3, C=128, 16x64x64, dilation=2, fp16 (1/4 of B=32)"),
}


class Workload:
    """Synthetic inputs of one configuration resident on `device` + closures for forward / backward
    through the MDCONV_CUDA entry points (the reference's positional signatures)."""

    def __init__(self, name, device, batch=None):
        import torch
        from modulated_deform_conv_amd import MDCONV_CUDA as M
        cfg = dict(WORKLOADS[name])
        self.name, self.cfg = name, cfg
        nd, B = cfg["nd"], batch or cfg["B"]
        C, O, sp, G, DG = cfg["C"], cfg["O"], cfg["sp"], cfg["G"], cfg["DG"]
        K = 3 ** nd
        self.B, self.K = B, K
        self.scale = B / cfg["B"]                       # work relative to the nominal shard
        self.n_samples = B * C * K * math.prod(sp)
        self.gemm_flop = 2.0 * O * (C // G) * K * B * math.prod(sp)   # ONE of the three GEMMs
        self.bytes = cfg["bytes"] * self.scale
        dt = {"f32": torch.float32, "f16": torch.float16}[cfg["dtype"]]
        g = torch.Generator(device="cpu").manual_seed(0)
        rn = lambda *s: torch.randn(*s, generator=g)
        mv = lambda t: t.to(device=device, dtype=dt).contiguous()
        self.x = mv(rn(B, C, *sp))
        self.off = mv(rn(B, DG * nd * K, *sp))
        self.m = mv(torch.sigmoid(rn(B, DG * K, *sp))) if cfg["modulated"] else None
        self.w = mv((torch.rand(O, C // G, *([3] * nd), generator=g) * 2 - 1) / math.sqrt(C * K))
        self.b = mv(0.1 * rn(O)) if cfg["bias"] else self.x.new_empty(0)
        self.go = mv(rn(B, O, *sp))
        d = cfg["dil"]
        self.geo = (3,) * nd + (1,) * nd + (d,) * nd + (d,) * nd + (G, DG, 64, cfg["bias"])
        self.M = M
        if not (nd == 2 and cfg["modulated"]):
            # caller-allocated entry points (reference deformable_conv3d.cu:160-167, 434-442 ...)
            self.out = torch.empty_like(self.go)
            self.gi, self.gw, self.gb = torch.empty_like(self.x), torch.empty_like(self.w), torch.empty_like(self.b)
            self.goff = torch.empty_like(self.off)
            self.gm = torch.empty_like(self.m) if self.m is not None else None

    def shard(self, lo, hi):
        for k in ("x", "off", "m", "go"):
            t = getattr(self, k)
            if t is not None:
                setattr(self, k, t[lo:hi].contiguous())
        self.scale *= (hi - lo) / self.B
        self.n_samples = self.n_samples // self.B * (hi - lo)
        self.gemm_flop *= (hi - lo) / self.B
        self.bytes *= (hi - lo) / self.B
        self.B = hi - lo

    def forward(self):
        M, c = self.M, self.cfg
        if c["nd"] == 2 and c["modulated"]:
            return M.modulated_deform_conv2d_forward_cuda(self.x, self.w, self.b, self.off, self.m, *self.geo)
        if c["modulated"]:
            M.modulated_deform_conv3d_forward_cuda(self.x, self.w, self.b, self.off, self.m, self.out, *self.geo)
        else:
            M.deform_conv3d_forward_cuda(self.x, self.w, self.b, self.off, self.out, *self.geo)
        return self.out

    def backward(self):
        """-> (grad_weight, grad_bias)"""
        from modulated_deform_conv_amd import _capi
        M, c = self.M, self.cfg
        if c["nd"] == 2 and c["modulated"]:
            r = M.modulated_deform_conv2d_backward_cuda(self.x, self.w, self.b, self.off, self.m, self.go, *self.geo)
            return r[3], r[4]
        with _capi.overwrite_grads():
            if c["modulated"]:
                M.modulated_deform_conv3d_backward_cuda(self.x, self.w, self.b, self.off, self.m, self.gi, self.gw,
                                                        self.gb, self.goff, self.gm, self.go, *self.geo)
            else:
                M.deform_conv3d_backward_cuda(self.x, self.w, self.b, self.off, self.gi, self.gw, self.gb,
                                              self.goff, self.go, *self.geo)
        return self.gw, self.gb


class _HipClock:
    """Device plumbing main() needs: synchronise, stamp, profile -- on the GPU."""
    stub = False
    dist_backend = "nccl"

    def setup(self, local_rank):
        import torch
        torch.cuda.set_device(local_rank)
        return torch.device("cuda", local_rank)

    def sync(self):
        import torch
        torch.cuda.synchronize()

    def stamp(self):
        import torch
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def elapsed_ms(self, a, b):
        return a.elapsed_time(b)

    def workload(self, name, device):
        return Workload(name, device)


class _StubWorkload:
    """HOST-ONLY stand-in for tests of the launcher / sharding / reporting logic (MDCONV_BENCH_STUB=1,
    tests/test_bench_cpu.py): same interface as Workload, a step that sleeps in proportion to its batch and returns
    CPU tensors for the exchange.  It computes nothing and is never used for a reported number."""

    def __init__(self, name, device):
        import torch
        cfg = dict(WORKLOADS[name])
        self.name, self.cfg, self.B = name, cfg, cfg["B"]
        self.K = 3 ** cfg["nd"]
        self.n_samples = self.B * cfg["C"] * self.K * math.prod(cfg["sp"])
        self.gemm_flop = 2.0 * cfg["O"] * (cfg["C"] // cfg["G"]) * self.K * self.B * math.prod(cfg["sp"])
        self.bytes, self.scale = cfg["bytes"], 1.0
        self.gw, self.gb = torch.ones(8, 8), torch.ones(8)

    def shard(self, lo, hi):
        f = (hi - lo) / self.B
        self.scale *= f
        self.n_samples = self.n_samples // self.B * (hi - lo)
        self.gemm_flop *= f
        self.bytes *= f
        self.B = hi - lo

    def forward(self):
        time.sleep(1e-4 * self.B)
        return self.gw

    def backward(self):
        time.sleep(2e-4 * self.B)
        return self.gw.clone(), self.gb.clone()


class _StubClock:
    stub = True
    dist_backend = "gloo"

    def setup(self, local_rank):
        import torch
        return torch.device("cpu")

    def sync(self):
        pass

    def stamp(self):
        return time.perf_counter()

    def elapsed_ms(self, a, b):
        return (b - a) * 1e3

    def workload(self, name, device):
        return _StubWorkload(name, device)


def resolve_world(gpus, env, visible_gpus):
    """How this invocation runs.  -> ("spawn", N): re-execute under torch.distributed.run with N ranks;
    ("run", world, rank, local_rank): run as that rank.  Raises SystemExit (non-zero) when the
    request cannot be met -- it never silently runs fewer ranks than `--gpus` asked for."""
    if gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "RANK" in env:
        world, rank = int(env.get("WORLD_SIZE", "1")), int(env["RANK"])
        local = int(env.get("LOCAL_RANK", "0"))
        if world != gpus:
            raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (gpus, world))
        if visible_gpus is not None and local >= visible_gpus:
            raise SystemExit("bench.py: rank %d needs GPU %d but only %d GPU(s) are visible" % (rank, local, visible_gpus))
        return ("run", world, rank, local)
    if visible_gpus is not None and visible_gpus < gpus:
        raise SystemExit("bench.py: --gpus %d requested but only %d GPU(s) are visible on this node; "
                         "refusing to print a mislabeled %d-GPU line" % (gpus, visible_gpus, visible_gpus))
    if gpus == 1:
        return ("run", 1, 0, 0)
    return ("spawn", gpus)


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


def _pmc_summary():
    """(data, source note) of the newest committed rocprofv3 --pmc summary (profiles/rNN*_pmc_summary.json, produced by
    tools/summarize_profile.py), or (None, note) when there is none or when it was collected on other kernel sources:
    counters cannot be collected inside this process, so the summary is stamped with the hash of the sources it was
    collected on and REFUSED when that differs from the sources being timed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not files:
        return None, None
    data = json.load(open(files[-1]))
    stamp = data.get("_kernel_sources_sha16")
    src = "%s (sources %s)" % (os.path.basename(files[-1]), stamp)
    if stamp != kernel_sources_sha16():
        return None, src + " -- stale: kernel sources changed since, traffic withheld"
    return data, src


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` (headline workload) from the committed PMC summary."""
    data, src = _pmc_summary()
    if data is None:
        return None, src
    for k, v in data.items():
        if k.startswith(kernel) and isinstance(v, dict):
            return v["hbm_bytes_per_launch"], src
    return None, src


def step_traffic(cfg, compulsory_bytes):
    """Counter HBM traffic of one forward + backward step of `cfg` (all kernels; separate FETCH_SIZE / WRITE_SIZE passes of
    tools/collect_profiles.sh over tools/bench_configs.py) against the compulsory bytes: the wasted-traffic ratio.  The read
    side is quoted raw and doubled (MI355X_MICROARCH.md: FETCH_SIZE tallies the 128-byte requests of a stream at 64 B; for
    gather-dominated kernels the raw figure is the closer one)."""
    data, src = _pmc_summary()
    ent = None if data is None else data.get("_steps", {}).get(cfg)
    if ent is None:
        return {"hbm_bytes_per_step": None, "source": src}
    tot = 2 * ent["fetch_bytes_raw"] + ent["write_bytes"]
    return {"hbm_bytes_per_step": tot, "fetch_bytes_raw": ent["fetch_bytes_raw"], "write_bytes": ent["write_bytes"],
            "x_compulsory": round(tot / compulsory_bytes, 2),
            "x_compulsory_fetch_raw": round((ent["fetch_bytes_raw"] + ent["write_bytes"]) / compulsory_bytes, 2),
            "kernels": ent.get("kernels"), "source": src}


def gather_path(wl, prof):
    """The grad_input gather (`north_star`: "rocprof HBM GB/s (gather path)"): algorithmic bytes of the timed gather kernel
    -- every grad_col row and every scatter-list entry read once, its result written once -- over its HIP-event time in
    this run (it runs on the forked stream BESIDE GEMM-2 where the backward forks, so this is its rate under contention;
    DESIGN.md section 4.2 quotes the alone-time)."""
    name = next((k for k in prof if "col2im" in k), None)
    if name is None or prof[name][1] <= 0:
        return None
    c = wl.cfg
    nd, es = c["nd"], (4 if c["dtype"] == "f32" else 2)
    pix, K, C, DG = wl.B * math.prod(c["sp"]), wl.K, c["C"], c["DG"]
    hp = c["dtype"] != "f32"
    cp = (C + 31) // 32 * 32 if hp else C
    rows = pix * K * cp * es                                   # grad_col rows [b][tap][pix][c]
    entry = 16 if nd == 2 else 32                              # bytes per list entry; fp32 2-D lists: one per corner PAIR
    entries = pix * K * DG * entry * (2 if (nd == 2 and not hp) else 1)
    two_pass = nd == 3 or hp
    ns = 2 ** (nd - 1)
    out = pix * ns * cp * es if two_pass else wl.B * C * math.prod(c["sp"]) * es   # per-anchor partial sums / grad_input
    alg = rows + entries + out
    ms = prof[name][1]
    return {"kernel": name, "ms": round(ms, 4), "algorithmic_bytes": int(alg),
            "achieved_GBs": round(alg / (ms * 1e-3) / 1e9, 1), "peak_GBs": HBM_PEAK_GBS,
            "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "note": "grad_col rows + list entries read once, %s written once; timed beside GEMM-2 where the backward forks"
                    % ("per-anchor partial sums" if two_pass else "grad_input")}


def cpu_baseline(iters=3):
    """The oracle (CPU restatement of the reference, kind = "port") on a bounded sample of the
    same workload: cfg2 at its full B = 32, `iters` forward + backward passes, median reported.
    im2col and the three GEMMs use every host thread OpenMP provides; the per-sample gradient
    loop (the reference's atomic scatter, mdeformable_conv.cu:202-318) runs on ONE thread so that
    its accumulation order is the reference's sequential one."""
    import oracle
    oracle.build()
    wl = Workload("cfg2", "cpu")
    times = []
    for _ in range(iters):
        t0 = time.perf_counter()
        oracle.forward(oracle.MDCN2D, wl.x, wl.w, wl.b, wl.off, wl.m, 1, 1, 1, 1, 1, 64)
        oracle.backward(oracle.MDCN2D, wl.x, wl.w, wl.b, wl.off, wl.m, wl.go, 1, 1, 1, 1, 1, 64)
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    return {"value": wl.n_samples / dt / 1e9, "unit": "GSamples/s",
            "cores": oracle.num_threads(), "threads_gradient_loop": 1, "kind": "port",
            "sample": "cfg2 at B=%d, %d x (fwd+bwd), median %.1f s per iteration (all: %s); %d OpenMP "
                      "threads for im2col + GEMMs, 1 thread for the per-sample gradient loop"
                      % (wl.B, iters, dt, ", ".join("%.1f" % t for t in times), oracle.num_threads())}


def time_other_config(name, device, steps=5, warmup=2):
    """Forward / backward of one of the other BASELINE.json configurations (its per-GPU shard) in
    this process: HIP-event time of `steps` forward and `steps` backward passes, the in-library
    per-kernel events, and the whole-step fraction of the roofline SURVEY.md 8d assigns it."""
    import torch
    from modulated_deform_conv_amd import _capi
    wl = Workload(name, device)
    for _ in range(warmup):
        wl.forward(); wl.backward()
    torch.cuda.synchronize()
    _capi.profile_enable(True)
    _capi.profile_reset()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    for _ in range(steps):
        wl.forward()
    ev[1].record()
    for _ in range(steps):
        wl.backward()
    ev[2].record()
    torch.cuda.synchronize()
    _capi.profile_enable(False)
    prof = _capi.profile_read()
    f_ms, b_ms = ev[0].elapsed_time(ev[1]) / steps, ev[1].elapsed_time(ev[2]) / steps
    t = (f_ms + b_ms) * 1e-3
    c = wl.cfg
    if c["bound"] == "hbm":
        achieved, unit = wl.bytes / t / 1e9, "GB/s"
    else:
        achieved, unit = 3 * wl.gemm_flop / t / 1e12, "TFLOP/s"
    res = {"workload": "%s, B=%d per GPU, forward+backward" % (c["what"], wl.B), "dtype": c["dtype"],
           "fwd_ms": round(f_ms, 4), "bwd_ms": round(b_ms, 4), "ms_per_step": round(f_ms + b_ms, 4),
           "GSamples_per_s": round(wl.n_samples / t / 1e9, 2), "kernel_path": _capi.last_kernels(),
           "roofline": {"bound": c["bound"], "achieved": round(achieved, 2), "peak": c["peak"], "unit": unit,
                        "frac": round(achieved / c["peak"], 4), "scope": "whole step",
                        "compulsory_bytes": int(wl.bytes), "gemm_flop": 3 * wl.gemm_flop},
           "kernels_ms": {k: round(v[1], 4) for k, v in prof.items()},
           "traffic": step_traffic(name, wl.bytes), "gather_path": gather_path(wl, prof)}
    del wl
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--sustain-s", type=float, default=2.0,
                    help="after the K timed steps, keep stepping for about this long and report sustained_ms_per_step "
                         "(0 = skip); not part of `value`")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--graph", dest="graph", action="store_true", default=None,
                    help="replay the step from a HIP graph (default for --scaling strong)")
    ap.add_argument("--no-graph", dest="graph", action="store_false")
    args = ap.parse_args()

    import torch
    be = _StubClock() if os.environ.get("MDCONV_BENCH_STUB") == "1" else _HipClock()
    visible = None if be.stub else (torch.cuda.device_count() if torch.cuda.is_available() else 0)
    plan = resolve_world(args.gpus, os.environ, visible)
    if plan[0] == "spawn":
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(plan[1]),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)
    _, world, rank, local_rank = plan
    if not be.stub and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    use_graph = (args.graph if args.graph is not None else args.scaling == "strong") and not be.stub
    # under torchrun (RANK set) the collective path is exercised even with one process
    distributed = world > 1 or (os.environ.get("MDCONV_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    device = be.setup(local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
            os.environ["NCCL_DEBUG"] = "WARN"   # the image's default prints a banner on stdout
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")   # ... and RCCL's warnings go there too: stdout carries ONE line
        if be.stub:
            dist.init_process_group(be.dist_backend)
        else:
            # RCCL prints a version banner (ROCm version / Hostname / Librccl path) on STDOUT when the communicator is
            # created, whatever NCCL_DEBUG_FILE says: stdout is pointed at stderr until the first collective has run, so
            # that rank 0's stdout carries the ONE JSON line and nothing else
            sys.stdout.flush()
            saved_fd = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group(be.dist_backend, device_id=device)
                dist.barrier()
                torch.cuda.synchronize()
            finally:
                sys.stdout.flush()
                os.dup2(saved_fd, 1)
                os.close(saved_fd)

    from modulated_deform_conv_amd import _capi
    from modulated_deform_conv_amd.distributed import FusedGradAllReduce, shard_bounds
    wl = be.workload("cfg2", device)
    nominal_b = wl.B
    if args.scaling == "strong":
        # global batch 32: this rank's contiguous shard
        lo, hi = shard_bounds(wl.B, world, rank)
        if hi <= lo:
            raise SystemExit("--scaling strong needs world size <= %d" % wl.B)
        wl.shard(lo, hi)
    reducer = FusedGradAllReduce() if distributed else None

    def compute():
        out = wl.forward()
        gw, gb = wl.backward()
        return out, gw, gb

    def compute_and_exchange():
        out, gw, gb = compute()
        # RCCL all-reduce of the fused [grad_weight || grad_bias] buffer on a side stream, released as soon as
        # GEMM-2 / grad_bias are done (the library's weights-ready event), i.e. under the grad_input gather
        reducer.reduce_overlapped(gw, gb)
        return out, gw, gb

    captured = False
    graph, exchange = None, ("none (one rank)" if reducer is None else "eager: side stream under the grad_input gather")
    if use_graph:
        # capture forward + backward once (plain kernel sequences on the capturing stream, scratch from the graph's
        # private pool).  With peers the exchange is captured TOO: the communication stream joins the capture through
        # the weights-ready event, so the replayed all-reduce runs under the gather exactly as in eager mode (until
        # round 5 it was issued after the replay -- serial, on the 0.55 ms step of an 8-way strong-scaling shard).  If
        # the collective cannot be captured on this stack, the step is re-captured without it and the exchange follows
        # the replay; `exchange` in the JSON line says which.
        # Capture mode "thread_local": the process group's watchdog thread polls the events of earlier collectives with
        # hipEventQuery; under the default (global) capture mode such a call from ANY thread while this thread captures is an
        # error that terminates the process (seen on the GPU box: "operation not permitted when stream is capturing" from
        # ProcessGroupNCCL::Watchdog, profiles/r06_experiments.md 5).
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                (compute_and_exchange if reducer is not None else compute)()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if reducer is not None and os.environ.get("MDCONV_BENCH_EXCHANGE") != "after":
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    static = compute_and_exchange()
                captured, exchange = True, "captured in the graph: side stream under the grad_input gather"
            except Exception as e:   # noqa: BLE001 -- any capture failure: fall back, say so
                sys.stderr.write("bench.py: the collective could not be captured (%s); exchange after the replay\n" % (e,))
                torch.cuda.synchronize()
                graph = None
        if graph is None:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                static = compute()
            if reducer is not None:
                exchange = "after the graph replay (serial)"

    def step():
        if graph is not None:
            graph.replay()
            if reducer is not None and not captured:
                reducer(static[1], static[2])      # after the replay, on the same stream
            return
        if reducer is not None:
            compute_and_exchange()
        else:
            compute()

    for _ in range(args.warmup):
        step()
    be.sync()
    paths = "stub" if be.stub else _capi.last_path()
    if graph is None and not be.stub:
        _capi.profile_enable(True)
        _capi.profile_reset()
    if distributed:
        dist.barrier()
    be.sync()
    t0 = time.perf_counter()
    marks = [be.stamp()]
    for i in range(args.steps):
        step()
        marks.append(be.stamp())
    t_enq = time.perf_counter()
    be.sync()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if be.stub:
        prof, prof_src = {"stub_kernel": (args.steps, 1.0)}, "stub"
    elif graph is None:
        _capi.profile_enable(False)
        prof, prof_src = _capi.profile_read(), "HIP events around each kernel inside the timed region"
    else:
        # event records do not time anything inside a graph: per-kernel averages from eager steps
        _capi.profile_enable(True)
        _capi.profile_reset()
        for _ in range(5):
            compute()
        torch.cuda.synchronize()
        _capi.profile_enable(False)
        prof, prof_src = _capi.profile_read(), "HIP events around each kernel, 5 eager steps after the timed (graph) region"

    if distributed:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    # (after the MAX over ranks: every rank derives the same step count from the same `elapsed`, so the collectives
    # inside step() stay matched)
    # Sustained figure: the timed region above is K steps (67 ms at the defaults), short enough for box-to-box clock
    # and cache state to show; the same step for >= --sustain-s seconds, HIP-event timed, makes those visible (and
    # gives a GPU-utilisation sampler something to see).  Reported next to the headline, never instead of it.
    sustained = None
    if args.sustain_s > 0:
        n_sus = max(args.steps, int(args.sustain_s / max(elapsed / args.steps, 1e-6)))
        s0 = be.stamp()
        for _ in range(n_sus):
            step()
        s1 = be.stamp()
        be.sync()
        sustained = (be.elapsed_ms(s0, s1) / n_sus, n_sus)

    if rank != 0:
        if distributed:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    step_ms = sorted(be.elapsed_ms(marks[i], marks[i + 1]) for i in range(args.steps))
    global_b = nominal_b * world if args.scaling == "weak" else nominal_b
    value = global_b * (wl.n_samples // wl.B) / (elapsed / args.steps) / 1e9
    # dominant kernel = the profiled kernel with the largest measured average duration
    gemms = {k: v for k, v in prof.items() if "col2im" not in k}
    dom, (dom_n, dom_ms) = max(gemms.items(), key=lambda kv: kv[1][1])
    achieved = wl.gemm_flop / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    traffic, traffic_src = (None, "stub") if be.stub else measured_traffic(dom)
    comp = wl.bytes
    result = {
        "metric": "fwd+bwd GSamples/s, MDCN2d 3x3 C=256 56x56 B=32; %HBM roofline",
        "value": round(value, 3), "unit": "GSamples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "ms_per_step_median": round(step_ms[len(step_ms) // 2], 4),
        "host_enqueue_ms_per_step": round((t_enq - t0) / args.steps * 1e3, 4),
        "sustained_ms_per_step": None if sustained is None else round(sustained[0], 4),
        "sustained_steps": None if sustained is None else sustained[1],
        "launch_mode": "hip graph replay" if graph is not None else "eager (one Python call per entry point)",
        "scaling": args.scaling, "exchange": exchange,
        "exchange_mode": None if reducer is None else reducer.last_mode, "vs_baseline": None, "dtype": "f32",
        "data": "STUB -- host-only launcher test, no kernel ran, not a measurement" if be.stub else "synthetic",
        "config": {"workload": "%s, B=%d per GPU, forward+backward (BASELINE.json configs[1])" % (wl.cfg["what"], wl.B),
                   "global_batch": global_b, "parallelism": "dp%d batch-sharded" % world,
                   "kernel_path": paths},
        "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2),
                     "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                     "traffic_source": traffic_src,
                     "flop_per_launch": wl.gemm_flop, "avg_ms": round(dom_ms, 4), "launches": dom_n},
        "kernels_ms": {k: round(v[1], 4) for k, v in prof.items()}, "kernels_ms_source": prof_src,
        "hbm_roofline": {"compulsory_bytes": int(comp),
                         "achieved_GBs": round(comp / (ms_per_step * 1e-3) / 1e9, 1),
                         "peak_GBs": HBM_PEAK_GBS,
                         "frac": round(comp / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        "traffic": ({"hbm_bytes_per_step": None, "source": "stub" if be.stub else "counters are collected on the B = 32 shard only"}
                    if be.stub or wl.B != nominal_b else step_traffic("cfg2", comp)),
        "gather_path": None if be.stub else gather_path(wl, prof),
    }
    if world == 1 and not distributed and not args.no_other_configs and not be.stub:
        del wl
        torch.cuda.empty_cache()
        result["other_configs"] = {n: time_other_config(n, device) for n in ("cfg3", "cfg4", "cfg5")}
    if world == 1 and not args.no_cpu_baseline and not be.stub:
        result["cpu_baseline"] = cpu_baseline()
    print(json.dumps(result))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
