"""Independent pure-PyTorch (autograd) statement of N-D (modulated) deformable convolution.

Used only to cross-check the C oracle: written from the operator's definition (SURVEY.md
section 8a "shared semantics"), not from the oracle's code, with all gradients coming from
autograd.  For non-integral sampling positions it must agree with the oracle to fp64 round-off.
"""
import itertools
import math

import torch


def _tup(v, nd):
    return (v,) * nd if isinstance(v, int) else tuple(v)


def deform_conv_nd(input, offset, mask, weight, bias, stride=1, padding=0, dilation=1,
                   groups=1, dgroups=1):
    B, C = input.shape[:2]
    in_sz = tuple(input.shape[2:])
    nd = len(in_sz)
    O = weight.shape[0]
    ksz = tuple(weight.shape[2:])
    stride, padding, dilation = _tup(stride, nd), _tup(padding, nd), _tup(dilation, nd)
    out_sz = tuple((in_sz[a] + 2 * padding[a] - (dilation[a] * (ksz[a] - 1) + 1)) // stride[a] + 1
                   for a in range(nd))
    K = math.prod(ksz)
    S_o = math.prod(out_sz)
    S_i = math.prod(in_sz)
    dt, dev = input.dtype, input.device

    # base sampling grid: [K, nd, S_o]
    taps = list(itertools.product(*[range(k) for k in ksz]))          # tap = (i*kw + j)[*kl + k]
    outs = torch.stack(torch.meshgrid(*[torch.arange(n) for n in out_sz], indexing="ij"), 0)
    outs = outs.reshape(nd, S_o).to(dt)
    base = torch.empty(K, nd, S_o, dtype=dt)
    for t, tap in enumerate(taps):
        for a in range(nd):
            base[t, a] = outs[a] * stride[a] - padding[a] + tap[a] * dilation[a]
    base = base.to(dev)

    # offset channel = dg * nd * K + nd * tap + axis, axis order (h, w[, l]), written out channel by channel from
    # mdeformable_conv.cu:65, 71-72 / deformable_conv3d.cu:92-94, 100-103 (not as one reshape, so that this
    # statement and the oracle do not share a single reading of the layout; tests/analytic_pins.py pins both)
    offs = offset.reshape(B, dgroups * nd * K, S_o)
    p = torch.stack([torch.stack([torch.stack([base[tap, axis][None] + offs[:, dg * nd * K + nd * tap + axis]
                                               for axis in range(nd)], 1)
                                  for tap in range(K)], 1)
                     for dg in range(dgroups)], 1)                     # [B, DG, K, nd, S_o]
    low = torch.floor(p).detach()
    d = p - low
    low = low.long()

    x = input.reshape(B, dgroups, C // dgroups, S_i)
    sample = torch.zeros(B, dgroups, C // dgroups, K * S_o, dtype=dt, device=dev)
    for corner in itertools.product((0, 1), repeat=nd):
        w = torch.ones(B, dgroups, K, S_o, dtype=dt, device=dev)
        valid = torch.ones(B, dgroups, K, S_o, dtype=torch.bool, device=dev)
        idx = torch.zeros(B, dgroups, K, S_o, dtype=torch.long, device=dev)
        for a in range(nd):
            pos = low[:, :, :, a] + corner[a]
            w = w * (d[:, :, :, a] if corner[a] else 1 - d[:, :, :, a])
            valid = valid & (pos >= 0) & (pos <= in_sz[a] - 1)
            idx = idx * in_sz[a] + pos.clamp(0, in_sz[a] - 1)
        w = torch.where(valid, w, torch.zeros_like(w))
        g = torch.gather(x, 3, idx.reshape(B, dgroups, 1, K * S_o).expand(-1, -1, C // dgroups, -1))
        sample = sample + w.reshape(B, dgroups, 1, K * S_o) * g
    sample = sample.reshape(B, dgroups, C // dgroups, K, S_o)
    if mask is not None:
        # mask channel = dg * K + tap (mdeformable_conv.cu:66, 73)
        mflat = mask.reshape(B, dgroups * K, S_o)
        mk = torch.stack([torch.stack([mflat[:, dg * K + tap] for tap in range(K)], 1) for dg in range(dgroups)], 1)
        sample = sample * mk.reshape(B, dgroups, 1, K, S_o)
    col = sample.reshape(B, groups, C // groups, K, S_o)
    wg = weight.reshape(groups, O // groups, C // groups, K)
    out = torch.einsum("gock,bgcks->bgos", wg, col).reshape(B, O, *out_sz)
    if bias is not None:
        out = out + bias.reshape(1, O, *([1] * nd))
    return out
