#!/usr/bin/env python3
"""Developer aid: forward + backward time and kernel path of realistic deformable layers (ResNet-DCN stages, wide / narrow
channel counts, 2-D and 3-D, fp32 / fp16), to spot shapes that leave the matrix-core kernels or run far below their peers.
usage: python tools/realistic_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.cases import D2, D3, M2, M3, _c, make_inputs, out_size
from tests.util import run_product

SHAPES = []
for (C, O, hw, B) in [(64, 64, 56, 16), (128, 128, 28, 16), (256, 256, 14, 16), (512, 512, 7, 16), (512, 512, 14, 8),
                      (256, 256, 56, 8), (64, 256, 56, 8), (256, 64, 56, 8), (1024, 1024, 7, 8), (2048, 512, 7, 8),
                      (96, 96, 40, 8), (192, 192, 20, 8), (320, 320, 10, 8)]:
    for dg in (1, 4):
        if C % dg == 0:
            SHAPES.append(("mdcn2d", M2, B, C, O, (hw, hw), dg))
for (C, O, sz, B) in [(32, 32, (8, 28, 28), 2), (64, 64, (8, 28, 28), 2), (64, 128, (8, 14, 14), 4), (128, 128, (4, 14, 14), 4),
                      (256, 256, (4, 7, 7), 4), (16, 16, (16, 32, 32), 2)]:
    SHAPES.append(("mdcn3d", M3, B, C, O, sz, 1))
    SHAPES.append(("dcn3d", D3, B, C, O, sz, 1))

# --more: a second set (other batch sizes, DCN without mask, bf16 is chosen with --dtypes) used to compare two trees on one box
MORE = []
for (C, O, hw, B) in [(256, 256, 28, 8), (128, 128, 56, 8), (256, 256, 56, 2), (256, 256, 56, 32), (64, 64, 112, 4), (512, 512, 28, 4),
                      (256, 512, 28, 8), (512, 256, 14, 16), (128, 256, 28, 16), (256, 256, 7, 32), (32, 32, 112, 8),
                      (256, 256, 14, 64), (1024, 256, 14, 4), (256, 1024, 14, 4)]:
    for dg in (1, 2):
        MORE.append(("mdcn2d", M2, B, C, O, (hw, hw), dg))
    MORE.append(("dcn2d", D2, B, C, O, (hw, hw), 1))
for (C, O, sz, B) in [(64, 64, (16, 28, 28), 2), (128, 128, (8, 14, 14), 4), (32, 64, (16, 56, 56), 1), (128, 128, (8, 28, 28), 2),
                      (256, 256, (4, 14, 14), 2), (64, 64, (4, 56, 56), 2)]:
    MORE.append(("mdcn3d", M3, B, C, O, sz, 1))
    MORE.append(("dcn3d", D3, B, C, O, sz, 1))


def main():
    global SHAPES
    dts = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}
    dtypes = (torch.float32, torch.float16)
    if "--more" in sys.argv:
        SHAPES = MORE
    if "--dtypes" in sys.argv:
        dtypes = tuple(dts[x] for x in sys.argv[sys.argv.index("--dtypes") + 1].split(","))
    for dtype in dtypes:
        for name, op, B, C, O, sz, dg in SHAPES:
            case = _c(name, op, B, C, O, sz, 3, dgroups=dg, tier="medium", seed=1)
            t = make_inputs(case, dtype=dtype, device="cuda")
            for _ in range(2):
                _, _, p = run_product(case, t, "auto")
            torch.cuda.synchronize()
            times = []
            for _ in range(7):   # median of single steps: one-off stalls (allocator growth, first use of an instance) do not count
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run_product(case, t, "auto")
                e1.record(); torch.cuda.synchronize()
                times.append(e0.elapsed_time(e1))
            ms = sorted(times)[len(times) // 2]
            K = 9 if len(sz) == 2 else 27
            so = 1
            for v in out_size(case):
                so *= v
            gflop = 3 * 2.0 * B * so * K * C * O / 1e9
            flag = "" if p == ["mfma", "mfma"] else "   <-- " + str(p)
            print("%-7s %-5s B=%-2d C=%-4d O=%-4d %-14s DG=%d  %8.3f ms  %7.1f TFLOP/s%s" % (
                name, str(dtype).replace("torch.", ""), B, C, O, "x".join(map(str, sz)), dg, ms, gflop / ms, flag), flush=True)
            del t
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
