"""Seeded synthetic workloads shared by the oracle tests, the golden fixtures and the GPU parity
tests.  Input distributions follow SURVEY.md section 8d: input, offset, grad_output ~ N(0,1),
mask = sigmoid(N(0,1)), weight ~ U(+-1/sqrt(C_in*K)) (reference reset_parameters,
modulated_deform_conv.py:432-439), bias ~ 0.1*N(0,1).  Continuous offsets avoid quirk Q2.
"""
import math
import random

import torch

import oracle

D2, M2, D3, M3 = oracle.DCN2D, oracle.MDCN2D, oracle.DCN3D, oracle.MDCN3D


def _c(name, op, B, C, O, in_sz, k, stride=1, padding=1, dilation=1, groups=1, dgroups=1,
       in_step=64, bias=True, tier="small", seed=0, offset_scale=1.0):
    return dict(name=name, op=op, B=B, C=C, O=O, in_sz=tuple(in_sz), k=k, stride=stride,
                padding=padding, dilation=dilation, groups=groups, dgroups=dgroups,
                in_step=in_step, bias=bias, tier=tier, seed=seed, offset_scale=offset_scale)


CASES = [
    # BASELINE.json configs[0] exactly: DeformConv2d 3x3, C_in=C_out=4, 8x8, B=1
    _c("cfg1_dcn2d_c4_8x8_b1", D2, 1, 4, 4, (8, 8), 3, bias=False, seed=1),
    # small sweeps: every op x stride / dilation / groups / deformable groups / bias / in_step
    _c("dcn2d_s2_g2_dg2", D2, 2, 8, 6, (9, 7), 3, stride=2, groups=2, dgroups=2, in_step=1, seed=2),
    _c("dcn2d_k1_dil1", D2, 3, 6, 5, (6, 6), 1, padding=0, in_step=2, bias=False, seed=3),
    _c("dcn2d_dil2_dg4", D2, 2, 8, 8, (10, 9), 3, padding=2, dilation=2, dgroups=4, seed=4),
    _c("mdcn2d_basic", M2, 2, 8, 8, (8, 8), 3, seed=5),
    _c("mdcn2d_s2_g4_dg2", M2, 4, 8, 12, (11, 10), 3, stride=2, groups=4, dgroups=2, in_step=3, seed=6),
    _c("mdcn2d_dil2_nobias", M2, 2, 6, 4, (9, 9), 3, padding=2, dilation=2, dgroups=3, bias=False, seed=7),
    _c("mdcn2d_k2_asym", M2, 2, 4, 4, (7, 9), 2, padding=1, in_step=2, seed=8),
    _c("mdcn2d_big_offsets", M2, 2, 4, 4, (8, 8), 3, seed=9, offset_scale=4.0),
    _c("mdcn2d_rect_params", M2, 2, 4, 6, (9, 8), (3, 2), stride=(2, 1), padding=(1, 0), dilation=(1, 2), seed=10),
    _c("dcn3d_basic", D3, 2, 4, 4, (5, 6, 4), 3, seed=11),
    _c("dcn3d_s2_g2", D3, 2, 4, 6, (6, 5, 7), 3, stride=2, groups=2, dgroups=2, in_step=1, bias=False, seed=12),
    _c("dcn3d_k2_dil2", D3, 1, 4, 4, (6, 6, 6), 2, padding=1, dilation=2, seed=13),
    _c("mdcn3d_basic", M3, 2, 4, 4, (4, 6, 5), 3, seed=14),
    _c("mdcn3d_dil2_dg2", M3, 2, 4, 8, (6, 7, 6), 3, padding=2, dilation=2, dgroups=2, in_step=1, seed=15),
    _c("mdcn3d_g2_big_offsets", M3, 2, 8, 4, (5, 5, 5), 3, groups=2, dgroups=4, seed=16, offset_scale=3.0),
    # small but MFMA-eligible (>= 16 channels per group): the golden fixtures of the MFMA path
    _c("mfma_mdcn2d_c32_o48_9x10", M2, 2, 32, 48, (9, 10), 3, in_step=1, seed=17),
    _c("mfma_dcn3d_c16_o16_5x6x5", D3, 2, 16, 16, (5, 6, 5), 3, bias=False, seed=18),
    # conv groups / deformable groups on the MFMA backward (block-diagonal dense weight, per-group
    # coordinate gradients, CSR keyed by deformable group, grouped col2im gather)
    _c("mfma_dcn2d_g2_c32_o32", D2, 2, 32, 32, (9, 8), 3, groups=2, in_step=1, seed=31),
    _c("mfma_mdcn2d_g4_dg2_c128_o64", M2, 2, 128, 64, (12, 11), 3, groups=4, dgroups=2, tier="medium", seed=32),
    _c("mfma_mdcn2d_g8_dg2_c256_o32", M2, 2, 256, 32, (7, 7), 3, groups=8, dgroups=2, tier="medium", seed=33),
    _c("mfma_mdcn3d_g2_dg2_c128_o32", M3, 1, 128, 32, (5, 6, 5), 3, groups=2, dgroups=2, in_step=1, tier="medium", seed=34),
    _c("cfg3s_mdcn2d_c256_g32_dg4_10x10", M2, 2, 256, 256, (10, 10), 3, groups=32, dgroups=4, bias=False, tier="medium", seed=35),
    _c("mfma_dcn2d_dg4_c1024_o16", D2, 1, 1024, 16, (6, 6), 3, dgroups=4, bias=False, tier="medium", seed=36),
    # deformable groups the fp32 matrix-core backward does not tile (C_in/DG of 16, 24, 32, 48): DG independent
    # single-group slices (mfma_kernels.hip, split_backward) -- one conv group, a slice inside a conv group
    # (weight / grad_output slices), a slice made of whole conv groups, 3-D
    _c("mfma_split_mdcn2d_dg4_c128_o128", M2, 2, 128, 128, (14, 13), 3, dgroups=4, tier="medium", seed=71),
    _c("mfma_split_dcn2d_g2_dg4_c128_o64", D2, 3, 128, 64, (9, 11), 3, groups=2, dgroups=4, in_step=1, tier="medium", seed=72),
    _c("mfma_split_mdcn2d_g4_dg2_c96_o64", M2, 2, 96, 64, (10, 9), 3, groups=4, dgroups=2, bias=False, tier="medium", seed=73),
    _c("mfma_split_mdcn3d_dg2_c32_o32", M3, 2, 32, 32, (5, 6, 5), 3, dgroups=2, in_step=1, tier="medium", seed=74),
    _c("mfma_split_dcn3d_s2_dg2_c48_o24", D3, 1, 48, 24, (7, 6, 7), 3, stride=2, dgroups=2, tier="medium", seed=75),
    _c("mfma_split_mdcn2d_dg8_c128_o32", M2, 2, 128, 32, (9, 10), 3, dgroups=8, tier="medium", seed=76),
    _c("mfma_split_dcn3d_g2_dg8_c128_o32", D3, 1, 128, 32, (5, 5, 6), 3, groups=2, dgroups=8, in_step=1, tier="medium", seed=77),
    # ... and as ONE padded problem (round 6, pad_plan: every group widened to 32-channel stages forward, 64 / 128 / n x 256
    # channels backward; the default with one conv group): groups of 8 / 24 / 40 / 80 / 136 channels, 2-D / 3-D, 2 - 5 groups
    _c("mfma_pad_mdcn2d_dg4_c96_o64", M2, 2, 96, 64, (11, 10), 3, dgroups=4, tier="medium", seed=78),
    _c("mfma_pad_dcn2d_dg2_c16_o16", D2, 3, 16, 16, (12, 13), 3, dgroups=2, in_step=1, bias=False, tier="medium", seed=79),
    _c("mfma_pad_mdcn2d_dg5_c200_o48_s2", M2, 2, 200, 48, (13, 12), 3, stride=2, dgroups=5, tier="medium", seed=80),
    _c("mfma_pad_dcn3d_dg2_c160_o32", D3, 1, 160, 32, (4, 5, 6), 3, dgroups=2, tier="medium", seed=81),
    _c("mfma_pad_mdcn3d_dg3_c72_o40_dil2", M3, 2, 72, 40, (5, 6, 5), 3, padding=2, dilation=2, dgroups=3, in_step=1, tier="medium", seed=82),
    _c("mfma_pad_mdcn2d_dg2_c272_o32", M2, 1, 272, 32, (7, 8), 3, dgroups=2, bias=False, tier="medium", seed=83),
    # one deformable group, C_in not a multiple of 64, enough pixels (3-D: 2048, 2-D below 64 channels: 8192): the padded plan
    # again, so that the channels-last kernels take the shape (round 6, pad_channels_preferred)
    _c("mfma_padc_mdcn3d_c32_o32_2304px", M3, 2, 32, 32, (8, 12, 12), 3, tier="medium", seed=84),
    _c("mfma_padc_dcn3d_c48_o24_s2", D3, 1, 48, 24, (14, 24, 25), 3, stride=2, in_step=1, bias=False, tier="medium", seed=85),
    _c("mfma_padc_mdcn3d_c16_o16_dil2", M3, 1, 16, 16, (10, 15, 16), 3, padding=2, dilation=2, tier="medium", seed=86),
    _c("mfma_padc_mdcn3d_c96_o40", M3, 1, 96, 40, (6, 18, 20), 3, tier="medium", seed=87),
    _c("mfma_padc_mdcn2d_c48_o32_9408px", M2, 3, 48, 32, (56, 56), 3, tier="medium", seed=88),
    # fewer than 16 input or output channels and more than a few hundred pixels: output channels padded to 16, input channels to
    # 64 (3-D, narrow 2-D) or to a multiple of 8, instead of the shape-generic kernels (round 6, experiment log 23)
    _c("mfma_padt_mdcn3d_c8_o8_600px", M3, 1, 8, 8, (6, 10, 10), 3, tier="medium", seed=89),
    _c("mfma_padt_dcn3d_c64_o8_576px", D3, 1, 64, 8, (4, 12, 12), 3, in_step=1, tier="medium", seed=90),
    _c("mfma_padt_mdcn2d_c32_o8_s2", M2, 2, 32, 8, (35, 33), 3, stride=2, tier="medium", seed=91),
    _c("mfma_padt_mdcn2d_c20_o12_nobias", M2, 1, 20, 12, (24, 25), 3, bias=False, tier="medium", seed=92),
    _c("mfma_padt_dcn2d_c8_o5_9408px", D2, 3, 8, 5, (56, 56), 3, tier="medium", seed=93),
    _c("mfma_padt_dcn3d_c3_o5_k2", D3, 2, 3, 5, (7, 9, 10), 2, padding=0, tier="medium", seed=94),
    # what the fp32 kernels do not tile at all, from 512 output pixels: C_in that is not a multiple of 8 (backward), output channels
    # below 16 with several deformable groups, deformable groups of 4 channels (16x padding) -- the padded plan instead of the
    # shape-generic kernels (experiment log 24)
    _c("mfma_padn_mdcn2d_c100_o40", M2, 2, 100, 40, (18, 17), 3, tier="medium", seed=95),
    _c("mfma_padn_dcn3d_c20_o24", D3, 1, 20, 24, (6, 10, 10), 3, in_step=1, bias=False, tier="medium", seed=96),
    _c("mfma_padn_mdcn2d_c128_dg2_o4", M2, 1, 128, 4, (24, 25), 3, dgroups=2, tier="medium", seed=97),
    _c("mfma_padn_mdcn2d_c16_dg4_o24", M2, 2, 16, 24, (18, 17), 3, dgroups=4, tier="medium", seed=98),
    _c("mfma_padn_dcn3d_c24_dg2_o8", D3, 1, 24, 8, (6, 10, 10), 3, dgroups=2, tier="medium", seed=99),
    # conv groups whose per-group channel counts the fp32 kernels do not tile (C_in / G not a multiple of 8 or below 16, fewer than
    # 16 output channels per group): padded per conv group (experiment log 28)
    _c("mfma_padg_mdcn2d_g2_c100_o40", M2, 2, 100, 40, (18, 17), 3, groups=2, tier="medium", seed=100),
    _c("mfma_padg_dcn3d_g4_c32_o8", D3, 1, 32, 8, (6, 10, 10), 3, groups=4, in_step=1, tier="medium", seed=101),
    _c("mfma_padg_mdcn2d_g2_c64_o8_nobias", M2, 1, 64, 8, (24, 25), 3, groups=2, bias=False, tier="medium", seed=102),
    _c("mfma_padg_dcn2d_g3_c36_o36_s2", D2, 2, 36, 36, (35, 33), 3, stride=2, groups=3, tier="medium", seed=103),
    _c("mfma_padg_mdcn3d_g2_c24_o40_k2", M3, 2, 24, 40, (7, 9, 10), 2, padding=0, groups=2, tier="medium", seed=104),
    _c("mfma_padg_dcn3d_g2_c80_o32_2304px", D3, 2, 80, 32, (8, 12, 12), 3, groups=2, tier="medium", seed=105),   # 40 -> 64 per group: slab rule
    # conv groups AND deformable groups, nested: equal (4 x 4 groups of 25), deformable groups finer (2 conv x 4 deformable of 12),
    # conv groups finer (4 conv x 2 deformable: 20 per conv group, 40 per deformable group), 3-D with few outputs per group
    _c("mfma_padgd_mdcn2d_g4_dg4_c100_o40", M2, 2, 100, 40, (18, 17), 3, groups=4, dgroups=4, tier="medium", seed=106),
    _c("mfma_padgd_dcn2d_g2_dg4_c48_o32", D2, 2, 48, 32, (18, 17), 3, groups=2, dgroups=4, in_step=1, tier="medium", seed=107),
    _c("mfma_padgd_mdcn2d_g4_dg2_c80_o64_nobias", M2, 1, 80, 64, (24, 25), 3, groups=4, dgroups=2, bias=False, tier="medium", seed=108),
    _c("mfma_padgd_dcn3d_g2_dg2_c40_o8", D3, 1, 40, 8, (6, 10, 10), 3, groups=2, dgroups=2, tier="medium", seed=109),
    # more than 64 KB of dynamic LDS in GEMM-1 (C_out = 512) and several channel passes (C_in = 512)
    _c("mfma_mdcn2d_c256_o512_6x6", M2, 1, 256, 512, (6, 6), 3, bias=False, tier="medium", seed=37),
    _c("mfma_dcn2d_c512_o32_7x5", D2, 2, 512, 32, (7, 5), 3, tier="medium", seed=38),
    # grad_out tile too large for LDS with the natural wave split (C_in <= 64, wide C_out): GEMM-1 puts more waves along the
    # channels (zero-padded blocks) instead of leaving the shape to the shape-generic kernels (round 5, bwd_dims)
    _c("mfma_mdcn3d_c64_o256_lds", M3, 1, 64, 256, (5, 6, 5), 3, tier="medium", seed=39),
    _c("mfma_dcn3d_c32_o320_g1_lds", D3, 2, 32, 320, (4, 5, 6), 3, bias=False, tier="medium", seed=40),
    _c("mfma_dcn2d_c64_o640_lds", D2, 2, 64, 640, (9, 8), 3, in_step=1, tier="medium", seed=46),
    _c("mfma_mdcn3d_c128_dg2_o512_lds", M3, 1, 128, 512, (4, 5, 5), 3, dgroups=2, tier="medium", seed=47),
    # ... the same idle-wave instances behind the deformable-group split (C_in / DG = 32 and 16 with a wide C_out, bias)
    _c("mfma_mdcn3d_c64_dg2_o256_lds_bias", M3, 1, 64, 256, (5, 6, 5), 3, dgroups=2, tier="medium", seed=48),
    _c("mfma_dcn3d_c32_dg2_o256_lds_bias", D3, 2, 32, 256, (4, 5, 6), 3, dgroups=2, in_step=1, tier="medium", seed=49),
    # kernel shapes / strides the MFMA kernels must also get right: 1x1, 5x5 stride 2 (25 taps =
    # three tap groups), rectangular, large dilation, anisotropic 3-D
    _c("mfma_mdcn2d_k1_c64_o32", M2, 2, 64, 32, (9, 11), 1, padding=0, tier="medium", seed=41),
    _c("mfma_dcn2d_k5_s2_c32_o64", D2, 2, 32, 64, (17, 15), 5, stride=2, padding=2, tier="medium", seed=42),
    _c("mfma_mdcn2d_k3x1_dil3_c48_o40", M2, 2, 48, 40, (13, 12), (3, 1), padding=(3, 0), dilation=(3, 1),
       bias=False, tier="medium", seed=43),
    _c("mfma_dcn3d_k1x3x3_s2_c32_o32", D3, 1, 32, 32, (4, 9, 10), (1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1),
       tier="medium", seed=44),
    _c("mfma_mdcn2d_big_offsets_c32", M2, 2, 32, 32, (8, 9), 3, tier="medium", seed=45, offset_scale=5.0),
    # channels-last forward (mfma_fwd_cl.hip): 3-D, C_in/groups a multiple of 64, all three tiles
    _c("cl_dcn3d_c64_o64_6x7x6", D3, 2, 64, 64, (6, 7, 6), 3, tier="medium", seed=51),
    _c("cl_mdcn3d_c128_o128_dil2", M3, 1, 128, 128, (4, 6, 5), 3, padding=2, dilation=2, bias=False, tier="medium", seed=52),
    _c("cl_dcn3d_c64_o264_s2", D3, 2, 64, 264, (5, 5, 6), 3, stride=2, tier="medium", seed=53),
    _c("cl_mdcn3d_g2_c128_o64_k2", M3, 1, 128, 64, (5, 4, 6), 2, padding=1, groups=2, tier="medium", seed=54),
    # medium: down-scaled analogues of BASELINE.json configs[1..4] (same K / stride / dilation /
    # G : DG structure, channel counts that exercise the MFMA tiles incl. ragged edges)
    _c("cfg2s_mdcn2d_c64_28x28_b4", M2, 4, 64, 64, (28, 28), 3, tier="medium", seed=21),
    _c("cfg2s_mdcn2d_c48_o80_ragged", M2, 3, 48, 80, (19, 23), 3, in_step=1, tier="medium", seed=22),
    _c("cfg2s_dcn2d_c64_28x28_b4", D2, 4, 64, 64, (28, 28), 3, tier="medium", seed=23),
    _c("cfg3s_mdcn2d_g8_dg4", M2, 4, 64, 64, (20, 20), 3, groups=8, dgroups=4, tier="medium", seed=24),
    _c("cfg4s_dcn3d_c16_12cubed_b2", D3, 2, 16, 16, (12, 12, 12), 3, tier="medium", seed=25),
    _c("cfg5s_mdcn3d_c16_dil2", M3, 2, 16, 16, (6, 14, 14), 3, padding=2, dilation=2, tier="medium", seed=26),
]

CASE_BY_NAME = {c["name"]: c for c in CASES}


def ndim(case):
    return 3 if case["op"] in (D3, M3) else 2


def _tup(v, nd):
    return (v,) * nd if isinstance(v, int) else tuple(v)


def out_size(case):
    nd = ndim(case)
    k, s, p, d = (_tup(case[x], nd) for x in ("k", "stride", "padding", "dilation"))
    return tuple((case["in_sz"][a] + 2 * p[a] - (d[a] * (k[a] - 1) + 1)) // s[a] + 1 for a in range(nd))


def make_inputs(case, dtype=torch.float32, device="cpu"):
    """Deterministic inputs for a case (generated on CPU in fp64, then cast/moved)."""
    g = torch.Generator().manual_seed(1000 + case["seed"])
    nd = ndim(case)
    k = _tup(case["k"], nd)
    K = math.prod(k)
    B, C, O = case["B"], case["C"], case["O"]
    osz = out_size(case)
    modulated = case["op"] in (M2, M3)

    def rn(*shape):
        return torch.randn(*shape, generator=g, dtype=torch.float64)

    t = {}
    t["input"] = rn(B, C, *case["in_sz"])
    t["offset"] = rn(B, case["dgroups"] * nd * K, *osz) * case["offset_scale"]
    t["mask"] = torch.sigmoid(rn(B, case["dgroups"] * K, *osz)) if modulated else None
    stdv = 1.0 / math.sqrt(C * K)
    t["weight"] = (torch.rand(O, C // case["groups"], *k, generator=g, dtype=torch.float64) * 2 - 1) * stdv
    t["bias"] = 0.1 * rn(O) if case["bias"] else None
    t["grad_output"] = rn(B, O, *osz)
    return {n: (None if v is None else v.to(dtype=dtype, device=device).contiguous()) for n, v in t.items()}


# ---- wide geometry: tap counts, strides and dilations beyond the case list (tools/fuzz_more.py --wide, tests/test_gpu_fuzz.py) ----
def _wide_geometry(r, nd, kmax, hi):
    """Per-axis kernel extents up to kmax (tap counts the case list and the other campaigns never reach: K = 16 ... 49 in 2-D,
    up to 125 in 3-D), strides and dilations up to 3, paddings up to 'same' + 1, input extents from the smallest that still gives
    one output position."""
    k = tuple(r.choice([1, 2, 3, 4, 5, kmax]) for _ in range(nd))
    if r.random() < 0.5:
        k = (max(k),) * nd
    stride = tuple(r.choice([1, 1, 2, 3]) for _ in range(nd))
    dil = tuple(r.choice([1, 1, 2, 3]) for _ in range(nd))
    pad = tuple(r.choice([0, 1, d_ * (k_ - 1) // 2, d_ * (k_ - 1) // 2 + 1]) for k_, d_ in zip(k, dil))
    size = tuple(r.randint(max(1, d_ * (k_ - 1) + 1 - 2 * p_), max(hi, d_ * (k_ - 1) + 2))
                 for k_, d_, p_ in zip(k, dil, pad))
    return k, stride, dil, pad, size


def case_f32_wide(seed):
    r = random.Random(99000 + seed)
    nd = r.choice([2, 2, 3])
    modulated = r.random() < 0.6
    op = {(2, False): D2, (2, True): M2, (3, False): D3, (3, True): M3}[(nd, modulated)]
    k, stride, dil, pad, size = _wide_geometry(r, nd, 7 if nd == 2 else 5, 22 if nd == 2 else 8)
    groups = r.choice([1, 1, 2])
    dg = r.choice([1, 1, 2, 4])
    C = dg * groups * r.choice([8, 16, 32]) if dg * groups > 1 else r.choice([16, 24, 32, 64, 128])
    O = r.choice([16, 17, 32, 48, 64, 128])
    O = (O + groups - 1) // groups * groups
    B = r.choice([1, 2, 3, 7, 17]) if nd == 2 else r.choice([1, 2, 5])
    return _c("wide%d" % seed, op, B, C, O, size, k, stride=stride, padding=pad, dilation=dil, groups=groups, dgroups=dg,
              in_step=r.choice([1, 64]), bias=r.random() < 0.5, tier="medium", seed=9000 + seed,
              offset_scale=r.choice([0.5, 2.0, 8.0]))


def case_hp_wide(seed):
    r = random.Random(66000 + seed)
    nd = r.choice([2, 2, 3])
    modulated = r.random() < 0.6
    op = {(2, False): D2, (2, True): M2, (3, False): D3, (3, True): M3}[(nd, modulated)]
    k, stride, dil, pad, size = _wide_geometry(r, nd, 7 if nd == 2 else 4, 16 if nd == 2 else 6)
    size = size[:-1] + (max(size[-1], 2),)
    groups = r.choice([1, 1, 2, 4])
    dg = r.choice([1, 1, 2, 4])
    C = r.choice([32, 64, 128])
    while C % groups or C % dg:
        C *= 2
    O = r.choice([8, 32, 48, 64, 128])
    O = (O + groups - 1) // groups * groups
    return _c("widehp%d" % seed, op, r.choice([1, 2, 3]), C, O, size, k, stride=stride, padding=pad, dilation=dil,
              groups=groups, dgroups=dg, in_step=64, bias=r.random() < 0.5, tier="medium", seed=6000 + seed,
              offset_scale=r.choice([0.5, 2.0, 8.0]))


# ---- shapes at the edges of the supported range (tools/extremes.py, tests/test_gpu_extremes.py; profiles/r05_experiments.md 26) ----
EXTREME_F32 = [
    # many taps
    _c("x_k11_2d", M2, 2, 16, 16, (20, 24), 11, padding=5, seed=1),
    _c("x_k31_2d", D2, 1, 16, 16, (36, 33), 31, padding=0, seed=2),
    _c("x_k35_2d_1225taps", M2, 1, 16, 16, (36, 38), 35, padding=0, seed=3, bias=False),   # generic backward: <= 1280 taps
    _c("x_k9_3d_729taps", M3, 1, 16, 16, (9, 10, 11), 9, padding=0, seed=4),
    _c("x_k7_3d_343taps_c64", D3, 2, 64, 64, (8, 8, 8), 7, padding=3, seed=5),
    _c("x_k1x49_2d", M2, 2, 32, 32, (5, 60), (1, 49), padding=(0, 0), seed=6),
    # long axes / many pixels
    _c("x_2d_1x65536", M2, 1, 16, 16, (1, 65536), (1, 3), padding=(0, 1), seed=7),
    _c("x_2d_3000x3000_c16", D2, 1, 16, 16, (3000, 3000), 3, seed=8, bias=False),
    _c("x_3d_256x4x4", M3, 1, 32, 32, (256, 4, 4), 3, seed=9),
    _c("x_2d_stride7_dil9", M2, 2, 32, 32, (80, 80), 3, stride=7, dilation=9, padding=9, seed=10),
    # wide channel counts
    _c("x_c2048_o2048_8x8", M2, 2, 2048, 2048, (8, 8), 3, seed=11),
    _c("x_c4096_o16_6x6_dg64", M2, 1, 4096, 16, (6, 6), 3, dgroups=64, seed=12),
    _c("x_c16_o4096_6x6", D2, 1, 16, 4096, (6, 6), 3, seed=13),
    _c("x_c1024_g32_dg32_3d", M3, 1, 1024, 1024, (4, 4, 4), 3, groups=32, dgroups=32, seed=14),
    # many images of one output position
    _c("x_b4096_1x1out", M2, 4096, 32, 32, (3, 3), 3, padding=0, seed=15),
    _c("x_b70000_1x1_k1", D2, 70000, 16, 16, (1, 1), 1, padding=0, seed=16),
    _c("x_b513_3d_1out", M3, 513, 32, 48, (3, 3, 3), 3, padding=0, seed=17),
    # offsets far outside the image
    _c("x_offsets_100px", M2, 2, 64, 64, (20, 20), 3, seed=18, offset_scale=100.0),
    _c("x_offsets_1e6px", D2, 2, 64, 64, (20, 20), 3, seed=19, offset_scale=1e6),
]
EXTREME_HP = [
    _c("xh_k7_2d", M2, 2, 64, 64, (20, 20), 7, padding=3, seed=31),
    _c("xh_k5_3d_125taps", M3, 1, 128, 128, (6, 8, 8), 5, padding=2, seed=32),
    _c("xh_k1x25", M2, 2, 128, 64, (4, 40), (1, 25), padding=0, seed=33),
    _c("xh_c2048_7x7", M2, 2, 2048, 512, (7, 7), 3, seed=34),
    _c("xh_c1024_dg16_g4", M2, 1, 1024, 256, (10, 10), 3, groups=4, dgroups=16, seed=35),
    _c("xh_2d_1x32768", M2, 1, 64, 64, (1, 32768), (1, 3), padding=(0, 1), seed=36),
    _c("xh_2d_1500x1500_c32", D2, 1, 32, 32, (1500, 1500), 3, seed=37),
    _c("xh_b4096_1x1out", M2, 4096, 64, 64, (3, 3), 3, padding=0, seed=38),
    _c("xh_3d_stride3_dil3", D3, 2, 64, 64, (20, 20, 20), 3, stride=3, dilation=3, padding=3, seed=39),
    _c("xh_offsets_100px", M3, 1, 64, 64, (8, 8, 8), 3, seed=40, offset_scale=100.0),
]
