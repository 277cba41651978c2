#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ (run from the repo root: python tests/golden/make_golden.py).

The reference cannot be built or imported in this image, so these vectors are
produced by the CPU oracle in fp64 -- after the oracle has been pinned by tests/test_oracle.py --
and committed so that (a) any later change of the oracle is caught and (b) the GPU parity tests
also have oracle-independent data.  One file per case: inputs (fp32 values, stored exactly) and
fp64 expected output + gradients."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from tests.cases import CASE_BY_NAME, make_inputs  # noqa: E402

GOLDEN_CASES = ["cfg1_dcn2d_c4_8x8_b1", "dcn2d_s2_g2_dg2", "mdcn2d_s2_g4_dg2", "mdcn2d_big_offsets",
                "mdcn2d_rect_params", "dcn3d_s2_g2", "mdcn3d_basic", "mfma_mdcn2d_c32_o48_9x10",
                "mfma_dcn3d_c16_o16_5x6x5",
                # conv groups + deformable groups on the MFMA backward
                "mfma_dcn2d_g2_c32_o32", "mfma_mdcn2d_g8_dg2_c256_o32",
                # channels-last 3-D kernels
                "cl_mdcn3d_g2_c128_o64_k2"]
ONLY_NEW = "--new" in sys.argv   # write only fixtures that do not exist yet


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name in GOLDEN_CASES:
        if ONLY_NEW and os.path.exists(os.path.join(out_dir, name + ".pt")):
            continue
        case = CASE_BY_NAME[name]
        t32 = make_inputs(case, dtype=torch.float32)             # the exact fp32 inputs the tests use
        t = {k: (None if v is None else v.double()) for k, v in t32.items()}
        args = (case["stride"], case["padding"], case["dilation"], case["groups"], case["dgroups"], case["in_step"])
        out = oracle.forward(case["op"], t["input"], t["weight"], t["bias"], t["offset"], t["mask"], *args)
        g = oracle.backward(case["op"], t["input"], t["weight"], t["bias"], t["offset"], t["mask"],
                            t["grad_output"], *args)
        blob = {"case": {k: v for k, v in case.items()},
                "inputs": {k: v for k, v in t32.items() if v is not None},
                "expected": {"output": out, **{k: v for k, v in g.items() if v is not None}}}
        # medium case: keep the file small by storing expected values in fp32
        if case["tier"] != "small":
            blob["expected"] = {k: v.float() for k, v in blob["expected"].items()}
        torch.save(blob, os.path.join(out_dir, name + ".pt"))
        print(name, os.path.getsize(os.path.join(out_dir, name + ".pt")) // 1024, "KiB")


if __name__ == "__main__":
    main()
