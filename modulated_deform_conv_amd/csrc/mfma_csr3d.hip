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
// cursor atomics and entries in the fill pass (1.21 ms -> see profiles/), the gather about equal.
#include "mfma_kernels.hpp"
#include "mfma_tile.hpp"

namespace mdconv {

namespace {

constexpr int kRun3 = 8;                 // targets per run
constexpr int kOob3 = 0x7ffffff0;        // out-of-range buffer offset: loads give 0

int grid_for3(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

// entry = 2 x int4: (tap * S_o + pixel, w(col), w(col + 1), low weight axis 0),
//                   (high weight axis 0, low weight axis 1, high weight axis 1, 0); mask in the column weights
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
      const int pos = rowptr[(int64_t)seg * (S_e + 1) + sa.qa] + atomicAdd(cursor + (int64_t)seg * S_e + sa.qa, 1);
      int4 *e = entries + ((int64_t)seg * ((int64_t)g.K * g.S_o) + pos) * 2;
      e[0] = make_int4(tap * g.S_o + pix, __float_as_int(sa.wx), __float_as_int(sa.wy), __float_as_int(sa.rl[0]));
      e[1] = make_int4(__float_as_int(sa.rh[0]), __float_as_int(sa.rl[1]), __float_as_int(sa.rh[1]), 0);
    }
  }
}

// LPD lanes (4 channels each) follow one list; a wave walks 64 / LPD runs of kRun3 consecutive
// targets side by side; workgroup tile = 4 * (64 / LPD) runs.  Channel units of LPD * 4 channels
// (inside one deformable group) are processed one after the other.
template <int LPD>
__global__ __launch_bounds__(256) void col2im3d_kernel(Geom g, int S_e, const float *__restrict__ gcol,
                                                       const int *__restrict__ rowptr,
                                                       const int4 *__restrict__ entries,
                                                       float *__restrict__ grad_input) {
  constexpr int NR = 4;                    // anchor rows that reach a target
  constexpr int NQ = 64 / LPD, RUNS = 4 * NQ, QT = RUNS * kRun3;
  constexpr int CW = LPD * 4;              // channels per unit
  constexpr int UB = LPD < 8 ? LPD : 8;    // row loads in flight per step
  constexpr int TP = QT + 1;               // LDS pitch
  __shared__ float tile[CW * TP];
  const int qtiles = (g.S_i + QT - 1) / QT;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int b = bid / qtiles;
  const int q0 = (bid - b * qtiles) * QT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane / LPD, r = lane % LPD;
  const rsrc_t r_gc = make_rsrc(gcol + (size_t)b * g.K * g.S_o * g.C, (size_t)g.K * g.S_o * g.C * 4);
  const int cseg = g.DG == 1 ? g.C : g.Cdg;     // channels that share one list
  const int upd = (cseg + CW - 1) / CW;         // units per segment
  const int units = g.DG * upd;
  const int qs = q0 + (wave * NQ + j) * kRun3;
  const int W = g.in_sz[2];
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
    // coordinates of the target of the current step (first step: qs - 1, the carry-in column)
    int tc[3];
    {
      int rem = max(qs - 1, 0);
      tc[2] = rem % g.in_sz[2]; rem /= g.in_sz[2];
      tc[1] = rem % g.in_sz[1];
      tc[0] = rem / g.in_sz[1];
      if (qs - 1 < 0) tc[2] = -1;
    }
    float4 cur = make_float4(0.f, 0.f, 0.f, 0.f), nxt = cur;
    for (int step = 0; step <= kRun3; ++step) {
      const int a = qs - 1 + step;
      const bool on = a >= 0 && a < g.S_i;
#pragma unroll
      for (int s = 0; s < NR; ++s) {
        // anchor row s: extended low index = target + s_a on the two outer axes
        const int er = (tc[0] + ((s >> 1) & 1)) * (g.in_sz[1] + 1) + tc[1] + (s & 1);
        const int ea = er * W + tc[2];
        const int e0 = on ? rp[ea] : 0, e1 = on ? rp[ea + 1] : 0;
        for (int base = e0; __any(base < e1); base += LPD) {
          const int cnt = max(0, min(LPD, e1 - base));
          int src_m = 0;
          float fx_m = 0.f, fy_m = 0.f;   // weights 0, row 0 beyond the list
          if (r < cnt) {
            const int4 ea4 = ent[(int64_t)(base + r) * 2], eb4 = ent[(int64_t)(base + r) * 2 + 1];
            // target = low + 1 - s_a on axis a: s_a = 1 -> the low side, 0 -> the high side
            const float rw = (((s >> 1) & 1) ? __int_as_float(ea4.w) : __int_as_float(eb4.x)) *
                             ((s & 1) ? __int_as_float(eb4.y) : __int_as_float(eb4.z));
            src_m = ea4.x;
            fx_m = rw * __int_as_float(ea4.y);
            fy_m = rw * __int_as_float(ea4.z);
          }
#pragma unroll
          for (int u0 = 0; u0 < LPD; u0 += UB) {
            float4 v[UB];
            float wx[UB], wy[UB];
#pragma unroll
            for (int k = 0; k < UB; ++k) {
              const int src = __shfl(src_m, u0 + k, LPD);
              wx[k] = __shfl(fx_m, u0 + k, LPD);
              wy[k] = __shfl(fy_m, u0 + k, LPD);
              v[k] = buf_load4(r_gc, src * g.C * 4 + c_voff, 0);
            }
#pragma unroll
            for (int k = 0; k < UB; ++k) {
              cur.x = fmaf(wx[k], v[k].x, cur.x); cur.y = fmaf(wx[k], v[k].y, cur.y);
              cur.z = fmaf(wx[k], v[k].z, cur.z); cur.w = fmaf(wx[k], v[k].w, cur.w);
              nxt.x = fmaf(wy[k], v[k].x, nxt.x); nxt.y = fmaf(wy[k], v[k].y, nxt.y);
              nxt.z = fmaf(wy[k], v[k].z, nxt.z); nxt.w = fmaf(wy[k], v[k].w, nxt.w);
            }
          }
        }
      }
      if (step > 0 && chan_on) {
        float *tp = tile + (r * 4) * TP + (a - q0);
        tp[0] = cur.x; tp[TP] = cur.y; tp[2 * TP] = cur.z; tp[3 * TP] = cur.w;
      }
      cur = nxt;
      nxt = make_float4(0.f, 0.f, 0.f, 0.f);
      // next target along the flattened image
      if (++tc[2] == W) {
        tc[2] = 0;
        if (++tc[1] == g.in_sz[1]) { tc[1] = 0; ++tc[0]; }
      }
    }
    __syncthreads();
    // transpose out: consecutive threads -> consecutive q of one channel
    for (int x = threadIdx.x; x < CW * QT; x += 256) {
      const int cl = x / QT, ql = x - cl * QT;
      const int c = c_lo + cl, q = q0 + ql;
      if (c < c_end && q < g.S_i) {
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

int col2im3d_f32(const Geom &g, const BwdDims &bd, const Tensors &t, const float *gcol,
                 const int *rowptr, const void *entries, hipStream_t stream) {
  const int cseg = g.DG == 1 ? g.C : g.Cdg;
  const int lanes = (cseg + 3) / 4;
#define C2I3(LPD)                                                                                \
  do {                                                                                           \
    const int qt = 4 * (64 / LPD) * kRun3;                                                       \
    hipLaunchKernelGGL((col2im3d_kernel<LPD>), dim3(g.B * ((g.S_i + qt - 1) / qt)), dim3(256), 0, \
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
