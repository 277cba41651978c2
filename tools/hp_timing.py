#!/usr/bin/env python3
"""Developer aid: per-phase cycle breakdown of hp_bwd2_kernel (needs the HP_TIMING build variant:
python -c "from modulated_deform_conv_amd import _build; print(_build.build_variant('timing', ['-DHP_TIMING']))"
then MDCONV_LIB=.../libmdconv_hip_timing.so python tools/hp_timing.py cfg5)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modulated_deform_conv_amd import _capi
import tools.bench_configs as bc
L = _capi.lib()
buf = (ctypes.c_ulonglong * 16)()
name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
bc.run(name)
torch.cuda.synchronize()
L.mdconv_debug_timing(buf, 1)
bc.run(name)
torch.cuda.synchronize()
L.mdconv_debug_timing(buf, 1)
labels = ["P2 gemm1", "B2 wait", "P3 request+gstore", "P3 consume0", "P3 item1", "B3 wait", "P4", "loop"]
for w in range(2):
    tot = sum(buf[w * 8:w * 8 + 8])
    print("wave %d: " % w + "  ".join("%s %.1f%%" % (labels[i], 100.0 * buf[w * 8 + i] / max(tot, 1)) for i in range(8)), " total cycles %.3g" % tot)
