// mfma_csr3d.hip -- fp32, 3-D: inverted scatter map keyed by SAMPLE and the grad_input gather.
//
// Reference: the 8 atomicAdds per sample of deform_conv3d_gradient_gpu_kernel
// (deformable_conv3d.cu:340-379; modulated: mdeformable_conv3d.cu:340-386).  Here, as in the 2-D
// scheme of mfma_bwd_data.hip, the data-dependent scatter is inverted into per-target lists and
// grad_input is GATHERED from the grad_col rows -- but with ONE list entry per sample instead of
// one per corner pair (4 in 3-D): the entry is keyed by the sample's low corner in the extended
// anchor space of mdconv_common.hpp (SampleAnchor) and carries the two column weights and the
// low / high weights of the two outer axes.  A target (z, y, x) then collects from the 4 anchor
// rows (z-1 | z, y-1 | y) at columns x-1 and x; the walk along x keeps the `cur` / `nxt`
// accumulators of the 2-D scheme, so a grad_col row is still read once per anchor row it feeds.
// Measured at cfg4 against the pair-keyed lists: 4x fewer counting atomics inside GEMM-1, 4x fewer
// cursor atomics and entries in the fill pass (1.21 -> 0.22 ms).
#include "mfma_kernels.hpp"
#include "mfma_tile.hpp"

namespace mdconv {

namespace {

constexpr int kRun3 = 4;                 // consecutive x targets of a run (x 2 x 2 in y, z)
constexpr int kOob3 = 0x7ffffff0;        // out-of-range buffer offset: loads give 0

int grid_for3(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

// entry = 2 x int4: (tap * S_o + pixel, w(col), w(col + 1), low weight axis 0),
//                   (high weight axis 0, low weight axis 1, high weight axis 1, anchor); mask in the column weights
template <bool MOD>
__global__ __launch_bounds__(256) void csr_fill3d_kernel(Geom g, int S_e, const float *__restrict__ offset,
                                                         const float *__restrict__ mask,
                                                         int *__restrict__ cursor,
                                                         const int *__restrict__ rowptr,
                                                         int4 *__restrict__ entries) {
  constexpr int ND = 3;
  const int64_t total = (int64_t)g.B * g.DG * g.K * g.S_o;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int pix = (int)(i % g.S_o);
    const int tap = (int)((i / g.S_o) % g.K);
    const int seg = (int)(i / g.S_o / g.K);   // b * DG + dg
    int oc[ND], tcd[ND];
    out_coords<ND>(g, pix, oc);
    tap_coords<ND>(g, tap, tcd);
    float delta[ND];
    const int64_t ob = ((int64_t)seg * (ND * g.K) + ND * tap) * g.S_o + pix;
#pragma unroll
    for (int a = 0; a < ND; ++a) delta[a] = offset[ob + (int64_t)a * g.S_o];
    TapCoef<ND, float> tc;
    make_tap<ND, float>(g, oc, tcd, delta, true, tc);
    const float m = MOD ? mask[((int64_t)seg * g.K + tap) * g.S_o + pix] : 1.f;
    SampleAnchor<ND> sa;
    sample_anchor<ND>(g, tc, m, sa);
    if (sa.on) {
      const int pos = rowptr[(int64_t)seg * (S_e + 1) + sa.qa] + atomicSub(cursor + (int64_t)seg * S_e + sa.qa, 1) - 1;   // counters count down
      int4 *e = entries + ((int64_t)seg * ((int64_t)g.K * g.S_o) + pos) * 2;
      e[0] = make_int4(tap * g.S_o + pix, __float_as_int(sa.wx), __float_as_int(sa.wy), __float_as_int(sa.rl[0]));
      e[1] = make_int4(__float_as_int(sa.rh[0]), __float_as_int(sa.rl[1]), __float_as_int(sa.rh[1]), sa.qa);
    }
  }
}

// LPD lanes (4 channels each) follow one list.  A RUN owns a 2 x 2 x kRun3 block of targets
// (z0 | z0+1, y0 | y0+1, kRun3 consecutive x) and walks the 3 x 3 anchor rows that reach it along
// x with the `cur` / `nxt` carry, so a grad_col row (and its list entry) is read ONCE for all the
// targets of the block it feeds: 9 anchor-row visits per 4 target rows instead of 16 (a single
// target row per run re-read every row 4x from HBM: L2 hit 27 %, 6.2 GB at cfg4).  A wave walks
// 64 / LPD runs side by side; workgroup tile = 4 * (64 / LPD) runs; channel units of LPD * 4
// channels (inside one deformable group) are processed one after the other.
// (cfg4: 0.91 -> 0.78 ms.  The same walk on the 16-bit rows of the cfg5 shard -- 203 registers with
// 8 channels per lane -- measured 7 % SLOWER than one target row per run and is not used there.)
template <int LPD>
__global__ __launch_bounds__(256) void col2im3d_kernel(Geom g, int S_e, const float *__restrict__ gcol,
                                                       const int *__restrict__ rowptr,
                                                       const int4 *__restrict__ entries,
                                                       float *__restrict__ grad_input) {
  constexpr int NQ = 64 / LPD, RUNS = 4 * NQ, RT = 4 * kRun3, QT = RUNS * RT;   // targets per run / tile
  constexpr int CW = LPD * 4;              // channels per unit
  constexpr int UB = LPD < 8 ? LPD : 8;    // row loads in flight per step
  constexpr int TP = QT + 1;               // LDS pitch
  __shared__ float tile[CW * TP];          // [channel][target row t][run][x]
  const int D = g.in_sz[0], H = g.in_sz[1], W = g.in_sz[2];
  const int nrx = (W + kRun3 - 1) / kRun3, nry = (H + 1) / 2, nrz = (D + 1) / 2;
  const int runs_img = nrx * nry * nrz, tiles_img = (runs_img + RUNS - 1) / RUNS;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int b = bid / tiles_img;
  const int r0 = (bid - b * tiles_img) * RUNS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane / LPD, r = lane % LPD;
  const rsrc_t r_gc = make_rsrc(gcol + (size_t)b * g.K * g.S_o * g.C, (size_t)g.K * g.S_o * g.C * 4);
  const int cseg = g.DG == 1 ? g.C : g.Cdg;     // channels that share one list
  const int upd = (cseg + CW - 1) / CW;         // units per segment
  const int units = g.DG * upd;
  const int run = wave * NQ + j, rg = r0 + run;
  const bool run_on = rg < runs_img;
  const int xs = (rg % nrx) * kRun3, y0 = ((rg / nrx) % nry) * 2, z0 = (rg / (nrx * nry)) * 2;
  for (int u = 0; u < units; ++u) {
    const int dg = u / upd;
    const int c_lo = dg * cseg + (u - dg * upd) * CW;          // first channel of the unit
    const int c_end = min(dg * cseg + cseg, g.C);              // end of the segment's channels
    const int c4 = c_lo + r * 4;
    const bool chan_on = c4 < c_end;
    const int seg = b * g.DG + dg;
    const int *rp = rowptr + (int64_t)seg * (S_e + 1);
    const int4 *ent = entries + (int64_t)seg * ((int64_t)g.K * g.S_o) * 2;
    const int c_voff = chan_on ? c4 * 4 : kOob3;
    float4 cur[4], nxt[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) cur[t] = nxt[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int step = 0; step <= kRun3; ++step) {
      const int xa = xs - 1 + step;          // anchor column: feeds targets xa (cur) and xa + 1 (nxt)
      const bool on = run_on && xa >= 0 && xa < W;
      // list bounds of the 9 anchor rows first (independent loads, one latency instead of nine).
      // Anchor row (extended low index) e = (z0 + dz, y0 + dy): on an axis it feeds target e - 1
      // with the sample's low weight and target e with its high weight
      int e0s[9], e1s[9];
#pragma unroll
      for (int dz = 0; dz < 3; ++dz)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int ez = z0 + dz, ey = y0 + dy;
          const bool row_on = on && ez <= D && ey <= H;
          const int ea = (ez * (H + 1) + ey) * W + xa;
          e0s[dz * 3 + dy] = row_on ? rp[ea] : 0;
          e1s[dz * 3 + dy] = row_on ? rp[ea + 1] : 0;
        }
#pragma unroll
      for (int dz = 0; dz < 3; ++dz) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int e0 = e0s[dz * 3 + dy], e1 = e1s[dz * 3 + dy];
          for (int base = e0; __any(base < e1); base += LPD) {
            const int cnt = max(0, min(LPD, e1 - base));
            int src_m = 0;
            float px_m[4], py_m[4];   // per target of the block: weight on column xa / xa + 1
#pragma unroll
            for (int t = 0; t < 4; ++t) px_m[t] = py_m[t] = 0.f;
            if (r < cnt) {
              const int4 ea4 = ent[(int64_t)(base + r) * 2], eb4 = ent[(int64_t)(base + r) * 2 + 1];
              src_m = ea4.x;
              const float fx = __int_as_float(ea4.y), fy = __int_as_float(ea4.z);
              const float zl = __int_as_float(ea4.w), zh = __int_as_float(eb4.x);
              const float yl = __int_as_float(eb4.y), yh = __int_as_float(eb4.z);
#pragma unroll
              for (int iz = 0; iz < 2; ++iz)
#pragma unroll
                for (int iy = 0; iy < 2; ++iy) {
                  if ((iz == dz - 1 || iz == dz) && (iy == dy - 1 || iy == dy)) {
                    const float w = (iz == dz ? zh : zl) * (iy == dy ? yh : yl);
                    px_m[iz * 2 + iy] = w * fx;
                    py_m[iz * 2 + iy] = w * fy;
                  }
                }
            }
#pragma unroll
            for (int u0 = 0; u0 < LPD; u0 += UB) {
              float4 v[UB];
#pragma unroll
              for (int k = 0; k < UB; ++k) {
                const int src = __shfl(src_m, u0 + k, LPD);
                v[k] = buf_load4(r_gc, src * g.C * 4 + c_voff, 0);
              }
#pragma unroll
              for (int k = 0; k < UB; ++k) {
#pragma unroll
                for (int iz = 0; iz < 2; ++iz)
#pragma unroll
                  for (int iy = 0; iy < 2; ++iy) {
                    if ((iz == dz - 1 || iz == dz) && (iy == dy - 1 || iy == dy)) {
                      const int t = iz * 2 + iy;
                      const float wx = __shfl(px_m[t], u0 + k, LPD), wy = __shfl(py_m[t], u0 + k, LPD);
                      cur[t].x = fmaf(wx, v[k].x, cur[t].x); cur[t].y = fmaf(wx, v[k].y, cur[t].y);
                      cur[t].z = fmaf(wx, v[k].z, cur[t].z); cur[t].w = fmaf(wx, v[k].w, cur[t].w);
                      nxt[t].x = fmaf(wy, v[k].x, nxt[t].x); nxt[t].y = fmaf(wy, v[k].y, nxt[t].y);
                      nxt[t].z = fmaf(wy, v[k].z, nxt[t].z); nxt[t].w = fmaf(wy, v[k].w, nxt[t].w);
                    }
                  }
              }
            }
          }
        }
      }
      if (step > 0 && chan_on) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float *tp = tile + (r * 4) * TP + t * (RUNS * kRun3) + run * kRun3 + step - 1;
          tp[0] = cur[t].x; tp[TP] = cur[t].y; tp[2 * TP] = cur[t].z; tp[3 * TP] = cur[t].w;
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) { cur[t] = nxt[t]; nxt[t] = make_float4(0.f, 0.f, 0.f, 0.f); }
    }
    __syncthreads();
    // transpose out: consecutive threads -> consecutive x of one (channel, target row)
    for (int x = threadIdx.x; x < CW * QT; x += 256) {
      const int cl = x / QT, ql = x - cl * QT;
      const int t = ql / (RUNS * kRun3), wr = (ql / kRun3) % RUNS, wk = ql % kRun3;
      const int wg = r0 + wr;
      const int wx = (wg % nrx) * kRun3 + wk, wy = ((wg / nrx) % nry) * 2 + (t & 1), wz = (wg / (nrx * nry)) * 2 + (t >> 1);
      const int c = c_lo + cl;
      if (c < c_end && wg < runs_img && wz < D && wy < H && wx < W) {
        float *dst = grad_input + ((int64_t)b * g.C + c) * g.S_i + (wz * H + wy) * W + wx;
        const float v = tile[cl * TP + ql];
        *dst = g.acc_data ? *dst + v : v;
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// Two-pass gather (round 3; the scheme of hp_col2im.hip in fp32): every grad_col row and every list
// entry is read ONCE.  The block walk above still fetches a row 9/4 times -- 2.5 GB (raw FETCH_SIZE) for
// the 1.8 GB of rows at cfg4, L2 hit 7 %.  Pass 1 walks ANCHOR rows and keeps, per anchor column, one
// partial sum for each of the 4 target rows (z - 1 | z, y - 1 | y) an anchor row feeds -- the same
// multiply-adds, spread over 4 accumulator sets -- written as fp32 rows A[segment][anchor][s][channels];
// pass 2 is a 4-point stencil over A (target t takes s from anchor row t + s) + the transpose to NCDHW.
// ---------------------------------------------------------------------------------------------
constexpr int kRunA3 = 16;   // anchors per run: the carry-in anchor is read twice (1 / 16 of the rows)

template <int LPD>
__global__ __launch_bounds__(256) void col2im3d_sums_kernel(Geom g, int S_e, const float *__restrict__ gcol,
                                                            const int *__restrict__ rowptr,
                                                            const int4 *__restrict__ entries,
                                                            float *__restrict__ sums) {
  constexpr int NS = 4, NQ = 64 / LPD, RUNS = 4 * NQ;
  constexpr int UB = 4;    // rows per load group; two groups are in flight
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane / LPD, r = lane % LPD;
  const int cseg = g.DG == 1 ? g.C : g.Cdg;     // channels that share one list
  const int runs_per_seg = (S_e + kRunA3 - 1) / kRunA3;
  const int blocks_per_seg = (runs_per_seg + RUNS - 1) / RUNS;
  const int seg = blockIdx.x / blocks_per_seg;                  // b * DG + dg
  const int run = (blockIdx.x - seg * blocks_per_seg) * RUNS + wave * NQ + j;
  const int b = seg / g.DG, dg = seg - b * g.DG;
  const bool run_on = run * kRunA3 < S_e;
  const int a_lo = run_on ? run * kRunA3 : 0;                    // first anchor this run writes
  const int a_last = run_on ? min(a_lo + kRunA3, S_e) - 1 : -1;  // last one
  const rsrc_t r_gc = make_rsrc(gcol + (size_t)b * g.K * g.S_o * g.C, (size_t)g.K * g.S_o * g.C * 4);
  const int *rp = rowptr + (int64_t)seg * (S_e + 1);
  const int4 *ent = entries + (int64_t)seg * ((int64_t)g.K * g.S_o) * 2;
  // channel units of LPD * 4 channels (more than 256 channels per list: the lists are walked once per unit)
  for (int cu = 0; cu < cseg; cu += LPD * 4) {
    const bool chan_on = cu + r * 4 < cseg;
    const int c_voff = chan_on ? (dg * cseg + cu + r * 4) * 4 : kOob3;
    float *out = sums + ((int64_t)seg * S_e * NS) * cseg + cu + r * 4;
    // The lists of consecutive anchors are contiguous in `entries`: a run streams ONE entry range, from
    // the carry-in anchor a_lo - 1 (only its column + 1 part lands here) to a_last, in batches of LPD
    // entries, rows loaded UB at a time with two groups in flight; an entry names its anchor and the
    // accumulators are flushed whenever the anchor advances (empty anchors included).
    const int e_end = run_on ? rp[a_last + 1] : 0;
    int e_pos = run_on ? rp[max(a_lo - 1, 0)] : 0;
    int cur_a = a_lo - 1;
    float4 cur[NS], nxt[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) cur[s] = nxt[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto flush = [&]() {   // anchor cur_a is complete: write its sums (not for the carry-in anchor), shift the column carry
      if (cur_a >= a_lo && chan_on) {
#pragma unroll
        for (int s = 0; s < NS; ++s) *reinterpret_cast<float4 *>(out + ((int64_t)cur_a * NS + s) * cseg) = cur[s];
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) { cur[s] = nxt[s]; nxt[s] = make_float4(0.f, 0.f, 0.f, 0.f); }
      ++cur_a;
    };
    while (__any(e_pos < e_end)) {
      int src_m = 0, anc_m = 0x7fffffff;
      float wx_m = 0.f, wy_m = 0.f, fa_m[NS] = {0.f, 0.f, 0.f, 0.f};
      if (e_pos + r < e_end) {
        const int4 ea4 = ent[(int64_t)(e_pos + r) * 2], eb4 = ent[(int64_t)(e_pos + r) * 2 + 1];
        src_m = ea4.x;
        wx_m = __int_as_float(ea4.y); wy_m = __int_as_float(ea4.z);
        const float zl = __int_as_float(ea4.w), zh = __int_as_float(eb4.x);
        const float yl = __int_as_float(eb4.y), yh = __int_as_float(eb4.z);
        // s: bit 1 = the target one lower in z (low weight), bit 0 = one lower in y
        fa_m[0] = zh * yh; fa_m[1] = zh * yl; fa_m[2] = zl * yh; fa_m[3] = zl * yl;
        anc_m = eb4.w;
      }
      const int cnt = max(0, min(LPD, e_end - e_pos));
      float4 va[UB], vb[UB];
      auto load_group = [&](float4 (&v)[UB], int u0) {
#pragma unroll
        for (int k = 0; k < UB; ++k) {
          const int src = __shfl(src_m, u0 + k, LPD);
          v[k] = buf_load4(r_gc, src * g.C * 4 + c_voff, 0);
        }
      };
      auto use_group = [&](const float4 (&v)[UB], int u0) {
#pragma unroll
        for (int k = 0; k < UB; ++k) {
          const int anc = __shfl(anc_m, u0 + k, LPD);
          const float wx = __shfl(wx_m, u0 + k, LPD), wy = __shfl(wy_m, u0 + k, LPD);
          float fa[NS];
#pragma unroll
          for (int s = 0; s < NS; ++s) fa[s] = __shfl(fa_m[s], u0 + k, LPD);
          if (u0 + k < cnt) {
            while (cur_a < anc) flush();
#pragma unroll
            for (int s = 0; s < NS; ++s) {
              const float ax = fa[s] * wx, ay = fa[s] * wy;
              cur[s].x = fmaf(ax, v[k].x, cur[s].x); cur[s].y = fmaf(ax, v[k].y, cur[s].y);
              cur[s].z = fmaf(ax, v[k].z, cur[s].z); cur[s].w = fmaf(ax, v[k].w, cur[s].w);
              nxt[s].x = fmaf(ay, v[k].x, nxt[s].x); nxt[s].y = fmaf(ay, v[k].y, nxt[s].y);
              nxt[s].z = fmaf(ay, v[k].z, nxt[s].z); nxt[s].w = fmaf(ay, v[k].w, nxt[s].w);
            }
          }
        }
      };
      load_group(va, 0);
#pragma unroll 1
      for (int u0 = 0; u0 < LPD; u0 += 2 * UB) {
        if (!__any(u0 < cnt)) break;          // (wave-uniform) nothing left in the batch for any run of the wave
        if (LPD > UB) load_group(vb, u0 + UB);
        use_group(va, u0);
        if (LPD > UB) {
          if (u0 + 2 * UB < LPD) load_group(va, u0 + 2 * UB);
          use_group(vb, u0 + UB);
        }
      }
      e_pos += LPD;
    }
    while (cur_a <= a_last) flush();
  }
}

// pass 2: grad_input[b][c][t] (+)= sum_s A[segment(b, c)][anchor row t + s][x][s][c]; workgroup = 64
// consecutive targets x 64 channels, lanes = (target, channel quad), LDS transpose to [B, C, S_i]
__global__ __launch_bounds__(256) void col2im3d_combine_kernel(Geom g, int S_e, const float *__restrict__ sums,
                                                               float *__restrict__ grad_input) {
  constexpr int NS = 4, QT = 64, CW = 64, TP = QT + 1;
  __shared__ float tile[CW * TP];
  const int qtiles = (g.S_i + QT - 1) / QT;
  const int b = blockIdx.x / qtiles, q0 = (blockIdx.x - b * qtiles) * QT;
  const int cseg = g.DG == 1 ? g.C : g.Cdg;
  const int D = g.in_sz[0], H = g.in_sz[1], W = g.in_sz[2];
  (void)D;
  for (int c0 = 0; c0 < g.C; c0 += CW) {
    // 16 lanes per target (64 channels), 16 targets per pass
    for (int it = 0; it < QT / 16; ++it) {
      const int ql = it * 16 + (threadIdx.x >> 4), q = q0 + ql;
      const int c = c0 + (threadIdx.x & 15) * 4;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < g.S_i && c < g.C) {
        const int dg = g.DG == 1 ? 0 : c / cseg;
        const int cl = c - dg * cseg;
        const int x = q % W, y = (q / W) % H, z = q / (W * H);
        const float *base = sums + ((int64_t)(b * g.DG + dg) * S_e * NS) * cseg + cl;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const int er = (z + (s >> 1)) * (H + 1) + y + (s & 1);
          const float4 v = *reinterpret_cast<const float4 *>(base + (((int64_t)er * W + x) * NS + s) * cseg);
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
      }
      float *tp = tile + ((threadIdx.x & 15) * 4) * TP + ql;
      tp[0] = acc.x; tp[TP] = acc.y; tp[2 * TP] = acc.z; tp[3 * TP] = acc.w;
    }
    __syncthreads();
    for (int x = threadIdx.x; x < CW * QT; x += 256) {
      const int cl = x / QT, ql = x - cl * QT;
      const int c = c0 + cl, q = q0 + ql;
      if (c < g.C && q < g.S_i) {
        float *dst = grad_input + ((int64_t)b * g.C + c) * g.S_i + q;
        const float v = tile[cl * TP + ql];
        *dst = g.acc_data ? *dst + v : v;
      }
    }
    __syncthreads();
  }
}

}  // namespace

int csr_fill3d_f32(const Geom &g, const BwdDims &bd, const Tensors &t, int *cursor,
                   const int *rowptr, void *entries, hipStream_t stream) {
  const int64_t samples = (int64_t)g.B * g.DG * g.K * g.S_o;
  if (g.modulated)
    hipLaunchKernelGGL((csr_fill3d_kernel<true>), dim3(grid_for3(samples)), dim3(256), 0, stream, g, bd.S_e,
                       (const float *)t.offset, (const float *)t.mask, cursor, rowptr, (int4 *)entries);
  else
    hipLaunchKernelGGL((csr_fill3d_kernel<false>), dim3(grid_for3(samples)), dim3(256), 0, stream, g, bd.S_e,
                       (const float *)t.offset, (const float *)t.mask, cursor, rowptr, (int4 *)entries);
  return check_launch("csr_fill3d");
}

size_t col2im3d_sums_bytes(const Geom &g) {
  const size_t cseg = g.DG == 1 ? g.C : g.Cdg;
  return (size_t)g.B * g.DG * hp_anchor_space(g) * 4 * cseg * sizeof(float);
}

int col2im3d_f32(const Geom &g, const BwdDims &bd, const Tensors &t, const float *gcol,
                 const int *rowptr, const void *entries, float *sums, hipStream_t stream) {
  const int cseg = g.DG == 1 ? g.C : g.Cdg;
  const int lanes = (cseg + 3) / 4;
  if (bd.two_pass) {
    const int runs_per_seg = (bd.S_e + kRunA3 - 1) / kRunA3;
#define C2S3(LPD)                                                                                \
  do {                                                                                           \
    const int runs = 4 * (64 / LPD);                                                             \
    hipLaunchKernelGGL((col2im3d_sums_kernel<LPD>), dim3(g.B * g.DG * ((runs_per_seg + runs - 1) / runs)), \
                       dim3(256), 0, stream, g, bd.S_e, gcol, rowptr, (const int4 *)entries, sums); \
  } while (0)
    if (lanes <= 4) C2S3(4);
    else if (lanes <= 8) C2S3(8);
    else if (lanes <= 16) C2S3(16);
    else if (lanes <= 32) C2S3(32);
    else C2S3(64);
#undef C2S3
    int rc = check_launch("col2im3d_sums");
    if (rc) return rc;
    hipLaunchKernelGGL(col2im3d_combine_kernel, dim3(g.B * ((g.S_i + 63) / 64)), dim3(256), 0, stream, g, bd.S_e,
                       sums, (float *)t.grad_input);
    return check_launch("col2im3d_combine");
  }
#define C2I3(LPD)                                                                                \
  do {                                                                                           \
    const int runs = 4 * (64 / LPD);                                                             \
    const int runs_img = ((g.in_sz[2] + kRun3 - 1) / kRun3) * ((g.in_sz[1] + 1) / 2) * ((g.in_sz[0] + 1) / 2); \
    hipLaunchKernelGGL((col2im3d_kernel<LPD>), dim3(g.B * ((runs_img + runs - 1) / runs)), dim3(256), 0, \
                       stream, g, bd.S_e, gcol, rowptr, (const int4 *)entries,                   \
                       (float *)t.grad_input);                                                   \
  } while (0)
  if (lanes <= 4) C2I3(4);
  else if (lanes <= 8) C2I3(8);
  else if (lanes <= 16) C2I3(16);
  else if (lanes <= 32) C2I3(32);
  else C2I3(64);
#undef C2I3
  return check_launch("col2im3d");
}

}  // namespace mdconv
