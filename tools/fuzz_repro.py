#!/usr/bin/env python3
"""Re-run single 16-bit shapes of a tools/fuzz_more.py campaign (developer aid, GPU box) and say what kind of error a failure is:
per tensor the scaled / per-element error in fp16, bf16 and through the fp32 kernels (MDCONV_HP=0: only the final rounding is
16-bit), and the worst element with its neighbours' magnitude.  Rounding noise scales with the type's epsilon (bf16 = 8 x fp16)
and vanishes on the fp32 kernels; a logic error does neither.
usage: python tools/fuzz_repro.py [--wide] seed [seed ...]"""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    a = sys.argv[1:]
    wide = "--wide" in a
    seeds = [int(x) for x in a if x.lstrip("-").isdigit()]
    if os.environ.get("FUZZ_REPRO_CHILD") != "1":
        for hp in ("1", "0"):
            env = dict(os.environ, FUZZ_REPRO_CHILD="1", MDCONV_HP=hp)
            print("## MDCONV_HP=%s (%s)" % (hp, "native 16-bit kernels" if hp == "1" else "fp32 kernels on widened copies"), flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__)] + a, env=env)
        return
    import torch
    spec = importlib.util.spec_from_file_location("fuzz_more", os.path.join(ROOT, "tools", "fuzz_more.py"))
    fm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fm)
    from tests.cases import make_inputs
    from tests.util import elem_err, rel_err, run_oracle, run_product
    gen = fm.case_hp_wide if wide else fm.case_hp
    for seed in seeds:
        case = gen(seed)
        print(case["name"], {k: case[k] for k in ("op", "B", "C", "O", "in_sz", "k", "stride", "padding", "dilation", "groups", "dgroups")})
        for dtype in (torch.float16, torch.bfloat16):
            t = make_inputs(case, dtype=dtype, device="cuda")
            out, grads, p = run_product(case, t, "auto")
            want_out, want = run_oracle(case, {k: (None if v is None else v.float()) for k, v in t.items()}, torch.float32)
            line = "  %-9s %s" % (str(dtype).replace("torch.", ""), p)
            worst = None
            for k, v in [("output", out)] + sorted(grads.items()):
                w = want_out if k == "output" else want[k]
                if v is None or w is None:
                    continue
                pe = elem_err(v.float(), w)
                line += "  %s %.1e/%.1e" % (k.replace("grad_", "g"), rel_err(v.float(), w), pe)
                if worst is None or pe > worst[0]:
                    worst = (pe, k, v.float().cpu().double().flatten(), w.cpu().double().flatten())
            print(line)
            pe, k, g, w = worst
            i = int(((g - w).abs() / (w.pow(2).mean().sqrt() + w.abs())).argmax())
            print("     worst %s[%d]: got %.6g want %.6g  rms %.3g  max %.3g  nonzero %d of %d"
                  % (k, i, g[i], w[i], w.pow(2).mean().sqrt(), w.abs().max(), int((w != 0).sum()), w.numel()), flush=True)


if __name__ == "__main__":
    main()
