"""Hand-derived known answers at INTEGER sample positions (zero offsets), where the four reference
files differ.  Scenario = my_test.py's (reference my_test.py:5-24) extended to
every op: input = 1 (1 x 1 x 5^nd), offset = 0, mask = 1, weight = 1 (1 x 1 x 3^nd), bias = 0,
stride 1, padding 1, grad_output = 1.  grad_col = W^T grad_out = 1 for every sample, so:

  * output / grad_input  = number of taps whose sample point lies inside the image (per pixel /
    per input pixel); grad_weight[tap] = number of pixels whose tap lies inside; grad_bias = 5^nd;
    grad_mask[tap, pixel] = val = [sample inside] -- identical for all four files;
  * grad_offset: with d == 0 the `abs(d) > EPS` gates of deformable_conv.cu:254-261 (loads) and
    of the 3-D files (deformable_conv3d.cu:336-338, mdeformable_conv3d.cu:336-338: loads AND
    atomics) skip every high corner, so only v1 survives and every axis gets
    grad_offset = -v1 * dval = -[sample inside]  (deformable_conv.cu:281-283,
    deformable_conv3d.cu:380-385, mdeformable_conv3d.cu:386-391; unconditional);
    modulated-2D has no such gate and reads v2..v4: the result is the right-sided difference
    v_high - v_low (mdeformable_conv.cu:295-314), non-zero only where the high neighbour leaves the
    image, and only for -1 < p < size (range gate :295): sum |grad_offset| = 52.
"""
import itertools

import torch

import oracle


def scenario(op):
    nd = 3 if op in (oracle.DCN3D, oracle.MDCN3D) else 2
    K = 3 ** nd
    sp = (5,) * nd
    t = dict(input=torch.ones(1, 1, *sp), offset=torch.zeros(1, nd * K, *sp),
             mask=torch.ones(1, K, *sp) if op in (oracle.MDCN2D, oracle.MDCN3D) else None,
             weight=torch.ones(1, 1, *([3] * nd)), bias=torch.zeros(1), grad_output=torch.ones(1, 1, *sp))
    return nd, t


def expected(op):
    """Analytic results of the scenario, built from the in-bounds indicator alone."""
    nd = 3 if op in (oracle.DCN3D, oracle.MDCN3D) else 2
    K = 3 ** nd
    sp = (5,) * nd
    inside = torch.zeros(K, *sp)            # [tap][pixel]: sample point inside the image
    grad_input = torch.zeros(*sp)
    for tap, tcoord in enumerate(itertools.product(range(3), repeat=nd)):
        for pix in itertools.product(range(5), repeat=nd):
            p = tuple(pix[a] - 1 + tcoord[a] for a in range(nd))
            if all(0 <= p[a] <= 4 for a in range(nd)):
                inside[(tap,) + pix] = 1
                grad_input[p] += 1
    e = dict(output=inside.sum(0)[None, None], grad_input=grad_input[None, None],
             grad_weight=inside.flatten(1).sum(1).view(1, 1, *([3] * nd)),
             grad_bias=torch.tensor([float(5 ** nd)]))
    if op in (oracle.MDCN2D, oracle.MDCN3D):
        e["grad_mask"] = inside[None]
    if op == oracle.MDCN2D:
        # right-sided difference of an all-ones image: -1 where the high neighbour along the axis is
        # outside while the sample itself is inside, 0 elsewhere
        go = torch.zeros(1, 2 * K, 5, 5)
        for tap, (i, j) in enumerate(itertools.product(range(3), repeat=2)):
            for h, w in itertools.product(range(5), repeat=2):
                ph, pw = h - 1 + i, w - 1 + j
                if 0 <= ph <= 4 and 0 <= pw <= 4:
                    go[0, 2 * tap, h, w] = -1.0 if ph + 1 > 4 else 0.0
                    go[0, 2 * tap + 1, h, w] = -1.0 if pw + 1 > 4 else 0.0
        e["grad_offset"] = go
    else:
        e["grad_offset"] = -inside.repeat_interleave(nd, dim=0)[None]   # channel = nd*tap + axis
    return e
