#!/usr/bin/env python3
"""Developer aid: per-phase cycle breakdowns (s_memtime stamps between the phases of a kernel's main loop, summed per
wave, one atomic per wave at exit).  A stamp costs ~200 cycles and waits for the scalar-memory / LDS counter, so it
serialises what it separates: fine for phases of thousands of cycles (GEMM-1 iterations, the 16-bit kernels' stages),
misleading for the 2.7 k-cycle chunks of the fp32 forward (--fwd: the stamps double that kernel's time).
  --gemm2 (B2_TIMING, mfma_bwd_weight_cl.hip)   --fwd2 (F2_TIMING, hp_fwd2.hip)   --bwd3 (B3_TIMING, hp_bwd3.hip)
  --fwd (F1_TIMING, mfma_fwd.hip)               default: GEMM-1 (B1_TIMING, mfma_bwd_data.hip).  Needs the
B1_TIMING build variant:
  python -c "from modulated_deform_conv_amd import _build; print(_build.build_one_file_variant('b1t', 'mfma_bwd_data.hip', ['-DB1_TIMING']))"
  MDCONV_LIB=.../libmdconv_hip_b1t.so python tools/b1_timing.py cfg2 cfg4"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modulated_deform_conv_amd import _capi
import bench
L = _capi.lib()
buf = (ctypes.c_ulonglong * 12)()
labels = ["tap state", "-", "bookkeeping", "gather issue", "MFMA quads", "consume", "collect/finish/park",
          "last drain", "barrier (previous tile done)", "grad_out tile -> LDS", "barrier (tile complete)", "ga emission + grad_bias"]
gemm2 = "--gemm2" in sys.argv   # needs the B2_TIMING variant of mfma_bwd_weight_cl.hip instead
if gemm2:
    buf = (ctypes.c_ulonglong * 8)()
    labels = ["prologue", "commit", "barrier", "tab issue", "MFMAs", "A issue", "gather issue (waits for its tab entry)", "partial stores"]
fwd2 = "--fwd2" in sys.argv     # F2_TIMING variant of hp_fwd2.hip: the 16-bit forward (runs wl.forward)
if fwd2:
    buf = (ctypes.c_ulonglong * 8)()
    labels = ["state build + fetch", "gathers + interpolation", "weights -> LDS", "barrier", "weight load issue",
              "matrix phase", "epilogue", "-"]
fwd1 = "--fwd" in sys.argv      # F1_TIMING variant of mfma_fwd.hip: the fp32 forward (runs wl.forward)
if fwd1:
    buf = (ctypes.c_ulonglong * 8)()
    labels = ["prologue", "commit", "barrier", "A issue", "gather issue", "MFMAs", "epilogue", "-"]
bwd3 = "--bwd3" in sys.argv     # B3_TIMING variant of hp_bwd3.hip
if bwd3:
    buf = (ctypes.c_ulonglong * 8)()
    labels = ["state build + fetch", "matrix phase", "barrier B1", "gather phase", "finish", "barrier B2", "prologue", "-"]
read = L.mdconv_debug_timing_f1 if fwd1 else L.mdconv_debug_timing_b3 if bwd3 else L.mdconv_debug_timing_f2 if fwd2 else (L.mdconv_debug_timing_b2 if gemm2 else L.mdconv_debug_timing_b1)
for name in ([a for a in sys.argv[1:] if not a.startswith("--")] or ["cfg2"]):
    wl = bench.Workload(name, "cuda")
    run = wl.forward if (fwd2 or fwd1) else wl.backward
    run(); torch.cuda.synchronize()
    read(buf, 1)
    run(); torch.cuda.synchronize()
    read(buf, 1)
    tot = float(sum(buf))
    print(name + ": " + "  ".join("%s %.1f%%" % (labels[i], 100.0 * buf[i] / max(tot, 1.0)) for i in range(len(labels))),
          " (total %.3g wave-cycles)" % tot)
