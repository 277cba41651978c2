# hp_fwd2 on grids of fewer workgroups than CU slots: narrower output-channel rows (MB) = more, lighter workgroups
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -3
cat > /tmp/fwd_only.py <<'PY'
import sys, os, torch
sys.path.insert(0, os.getcwd())
from tools.prof_shape import parse
from tests.cases import make_inputs
from modulated_deform_conv_amd import MDCONV_CUDA as M
for spec in sys.argv[1:]:
    case, dtype = parse(spec)
    t = make_inputs(case, dtype=dtype, device="cuda")
    geo = (3, 3, 1, 1, 1, 1, 1, 1, case["groups"], case["dgroups"], 64, True)
    f = lambda: M.modulated_deform_conv2d_forward_cuda(t["input"], t["weight"], t["bias"], t["offset"], t["mask"], *geo)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print("%-40s fwd %.1f us" % (spec, e0.elapsed_time(e1) / 20 * 1e3), flush=True)
PY
S="m2:f16:B8:C256:O256:56x56:dg1 m2:f16:B4:C256:O256:56x56:dg1 m2:f16:B2:C256:O256:56x56:dg1 m2:f16:B16:C256:O256:28x28:dg1 m2:f16:B16:C256:O256:14x14:dg1 m2:f16:B32:C256:O256:56x56:dg1 m2:f16:B8:C128:O256:56x56:dg1"
for mb in 8 4 2 1; do echo "== MB <= $mb"; MDCONV_HP_FWD_MB=$mb python /tmp/fwd_only.py $S 2>&1 | grep -v amdgpu.ids; done
