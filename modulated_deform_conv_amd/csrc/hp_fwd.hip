// hp_fwd.hip -- forward for fp16 / bf16 tensors on v_mfma_f32_32x32x16_{f16,bf16} (gfx950).
//
//   out[o, n] = sum_{tap, c} W[o, c, tap] * ( mask[tap, n] * interp(input[c], p(tap, n)) )
//
// Reference: im2col kernel + per-group addmm_ (mdeformable_conv.cu:37-87, 172-183; 3-D:
// mdeformable_conv3d.cu:54-127, 230-245); dtype dispatch incl. half: mdeformable_conv.cu:101.
//
// M = output channels (A = weights, pre-packed in fragment order, staged through LDS and shared by
// the four waves), N = output pixels, K = (tap, input channel).  Every wave owns 32 pixels and ALL
// output channels of the workgroup (MB blocks of 32), so the column operand is needed by exactly
// one wave -- and a lane (pixel = lane & 31, channel octet = lane >> 5) that gathers 8 channels of
// its pixel's corners with 16-byte loads from the channels-last copy holds, after interpolation,
// precisely its B fragment of the 32x32x16 MFMA: the column operand never touches LDS.
// At 16x the fp32 matrix rate the kernel is bound by the gather path (64 B/clk/CU), not by the
// matrix cores, so everything else is arranged to stay out of the texture path's way:
// weights go global -> LDS once per workgroup and K stage (4 chunks of 16 channels), interpolation
// is one v_fma_mix_f32 per corner and channel (fp16 source, fp32 weight and accumulator).
// Conv groups: the packed weight is block diagonal; a per-chunk table says which output-channel
// blocks can be non-zero, the others are skipped (uniform branches) and never staged.
#include "hp_kernels.hpp"

namespace mdconv {

namespace {

constexpr int kStage = 4;   // 16-channel chunks per LDS stage of the weights

template <int ND, bool MOD, typename T, int MB>
__global__ __launch_bounds__(256, 2) void hp_fwd_kernel(
    Geom g, HpDims hd, const typename T::Raw *__restrict__ xt, const U4 *__restrict__ wpf,
    const typename T::Raw *__restrict__ bias, const typename T::Raw *__restrict__ offset,
    const typename T::Raw *__restrict__ mask, typename T::Raw *__restrict__ output,
    const int2 *__restrict__ ctab) {
  constexpr int NC = 1 << ND;
  __shared__ U4 As[2][kStage][MB][64];

  const int tid = threadIdx.x, lane = tid & 63, kh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int orange = blockIdx.y;
  const int ntiles = gridDim.x;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int nchunks = hd.Cp / 16;
  const int2 *ct = ctab + orange * (nchunks + 1);
  // chunk range of this workgroup's output channels (contiguous: conv groups are contiguous)
  const int2 rng = ct[nchunks];
  const int ch_lo = rng.x, ch_hi = rng.y;
  if (ch_lo >= ch_hi) return;

  // ---- this lane's output pixel ----
  const int n_raw = tile * 128 + wave * 32 + (lane & 31);
  const bool live = n_raw < g.N;
  const int n = live ? n_raw : g.N - 1;
  const int b = n / g.S_o, pix = n - b * g.S_o;
  int oc[ND];
  out_coords<ND>(g, pix, oc);

  const rsrc_t r_xt = make_rsrc(xt, (size_t)g.B * g.S_i * hd.Cp * 2);
  const rsrc_t r_w = make_rsrc(wpf, (size_t)g.K * nchunks * hd.oblks * 1024);
  const int img_off = b * g.S_i;

  f32x16 acc[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // ---- sampling state of the (tap, deformable group) being REQUESTED ----
  int voff[NC];
  float wgt[NC];
  int st_tap = -1, st_dg = -1;
  typename T::Raw dlr[ND], mlr;   // prefetched offsets / mask of (pf_tap, pf_dg), raw until used
  int pf_tap = -1, pf_dg = -1;
  auto fetch = [&](int tap, int dg) {
    const int64_t seg = (int64_t)b * g.DG + dg;
    const int64_t ob = (seg * (ND * g.K) + ND * tap) * g.S_o + pix;
#pragma unroll
    for (int a = 0; a < ND; ++a) dlr[a] = offset[ob + (int64_t)a * g.S_o];
    if (MOD) mlr = mask[(seg * g.K + tap) * g.S_o + pix];
    pf_tap = tap;
    pf_dg = dg;
  };
  auto build = [&](int tap, int dg) {
    if (pf_tap != tap || pf_dg != dg) fetch(tap, dg);
    int tcd[ND];
    tap_coords<ND>(g, tap, tcd);
    float dl[ND];
#pragma unroll
    for (int a = 0; a < ND; ++a) dl[a] = T::ldf(&dlr[a]);
    const float ml = MOD ? T::ldf(&mlr) : 1.f;
    TapCoef<ND, float> tc;
    make_tap<ND, float>(g, oc, tcd, dl, false, tc);
    HpCorners<ND> hc;
    hp_corners<ND>(tc, hc);
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) {
      voff[ci] = hc.idx[ci] >= 0 ? ((img_off + hc.idx[ci]) * hd.Cp + 8 * kh) * 2 : kHpOob;
      wgt[ci] = hc.w[ci] * ml;
    }
    st_tap = tap;
    st_dg = dg;
  };
  auto dg_of = [&](int ch) { return g.DG == 1 ? 0 : min(ch * 16, g.C - 1) / g.Cdg; };
  // the (tap, dg) that follows (tap, ch)'s in the walk, for the offset / mask prefetch
  auto next_state = [&](int tap, int ch, int &ntap, int &ndg) {
    const int dg = dg_of(ch);
    int c2 = ch;
    while (c2 < ch_hi && dg_of(c2) == dg) ++c2;
    if (c2 < ch_hi) { ntap = tap; ndg = dg_of(c2); }
    else { ntap = min(tap + 1, g.K - 1); ndg = dg_of(ch_lo); }
  };

  struct Set { U4 v[NC]; float w[NC]; };
  auto issue = [&](Set &s, int tap, int ch) {
    const int dg = dg_of(ch);
    if (tap != st_tap || dg != st_dg) {
      build(tap, dg);
      int ntap, ndg;
      next_state(tap, ch, ntap, ndg);
      fetch(ntap, ndg);
    }
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) {
      s.v[ci] = buf_load4u(r_xt, voff[ci], ch * 32);
      s.w[ci] = wgt[ci];
    }
  };

  // ---- weight staging: stage = (tap, kStage chunks); fragment f of a stage = (chunk f / MB,
  // block f % MB); wave w moves fragments w, w + 4, ... ----
  constexpr int FPW = (kStage * MB + 3) / 4;
  U4 wr[FPW];
  auto w_load = [&](int tap, int ch0) {
#pragma unroll
    for (int k = 0; k < FPW; ++k) {
      const int f = wave + 4 * k;
      const int j = f / MB, ob = f % MB;
      const int ch = ch0 + j;
      bool on = f < kStage * MB && ch < ch_hi;
      if (on) {
        const int2 e = ct[ch];
        on = ob >= e.x && ob < e.x + e.y;
      }
      if (on)
        wr[k] = buf_load4u(r_w, lane * 16, ((tap * nchunks + ch) * hd.oblks + orange * MB + ob) * 1024);
    }
  };
  auto w_store = [&](int buf) {
#pragma unroll
    for (int k = 0; k < FPW; ++k) {
      const int f = wave + 4 * k;
      if (f < kStage * MB) As[buf][f / MB][f % MB][lane] = wr[k];
    }
  };

  auto consume = [&](const Set &s, int ch, int buf) {
    float col[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) col[j] = 0.f;
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) mac8<T>(col, s.v[ci], s.w[ci]);
    const U4 bfrag = pack8<T>(col);
    const int2 e = ct[ch];
    const int j = (ch - ch_lo) % kStage;
#pragma unroll
    for (int ob = 0; ob < MB; ++ob) {
      if (ob >= e.x && ob < e.x + e.y) acc[ob] = T::mfma(As[buf][j][ob][lane], bfrag, acc[ob]);
    }
  };

  // ---- walk: taps x chunks, two gather register sets, weights one stage ahead ----
  Set sa, sb;
  int tap = 0, ch = ch_lo, stage = 0;
  int itap = 0, ich = ch_lo;   // position of the next step to request
  auto advance = [&](int &t, int &c) {
    if (++c == ch_hi) { c = ch_lo; ++t; }
  };
  w_load(0, ch_lo);
  issue(sa, itap, ich);
  advance(itap, ich);
  bool done = false;
  while (!done) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      Set &cur = half == 0 ? sa : sb;
      Set &nxt = half == 0 ? sb : sa;
      if (done) break;
      if ((ch - ch_lo) % kStage == 0) {
        // stage boundary: publish this stage's weights, request the next stage's
        w_store(stage & 1);
        __syncthreads();
        int t2 = tap, c2 = ch + kStage;
        if (c2 >= ch_hi) { c2 = ch_lo; ++t2; }
        if (t2 < g.K) w_load(t2, c2);
      }
      if (itap < g.K) issue(nxt, itap, ich);
      advance(itap, ich);
      consume(cur, ch, stage & 1);
      const int chn = ch + 1;
      if (chn == ch_hi || (chn - ch_lo) % kStage == 0) ++stage;
      advance(tap, ch);
      done = tap >= g.K;
    }
  }

  // ---- epilogue: + bias, store [B, O, S_o]; lanes 0-31 -> 32 consecutive pixels ----
  if (live) {
    if (g.with_bias) {   // bias values first, as independent loads (hp_fwd2.hip)
#pragma unroll
      for (int ob = 0; ob < MB; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = (orange * MB + ob) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
          acc[ob][r] += T::ldf(bias + (o < g.O ? o : 0));
        }
    }
#pragma unroll
    for (int ob = 0; ob < MB; ++ob)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = (orange * MB + ob) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (o < g.O) T::stf(output + ((int64_t)b * g.O + o) * g.S_o + pix, acc[ob][r]);
      }
  }
}

}  // namespace

template <int ND, bool MOD, typename T>
static int launch_fwd_hp(const Geom &g, const HpDims &hd, const Tensors &t, const void *xt,
                         const void *wpf, const int2 *ctab, hipStream_t stream) {
  using Raw = typename T::Raw;
  const dim3 grid((g.N + 127) / 128, hd.oranges);
#define HP_FWD(MBV)                                                                              \
  hipLaunchKernelGGL((hp_fwd_kernel<ND, MOD, T, MBV>), grid, dim3(256), 0, stream, g, hd,          \
                     (const Raw *)xt, (const U4 *)wpf, (const Raw *)t.bias, (const Raw *)t.offset, \
                     (const Raw *)t.mask, (Raw *)t.output, ctab)
  switch (hd.MB) {
    case 1: HP_FWD(1); break;
    case 2: HP_FWD(2); break;
    default: HP_FWD(4); break;   // (hp_dims: at most 4 output-channel blocks per row)
  }
#undef HP_FWD
  return check_launch("hp_fwd");
}

int hp_forward_launch(const Geom &g, const HpDims &hd, int dtype, const Tensors &t, const void *xt,
                      const void *wpf, const int2 *ctab, hipStream_t stream) {
#define HP_DISPATCH(T)                                                                       \
  do {                                                                                       \
    if (g.nd == 2)                                                                           \
      return g.modulated ? launch_fwd_hp<2, true, T>(g, hd, t, xt, wpf, ctab, stream)         \
                         : launch_fwd_hp<2, false, T>(g, hd, t, xt, wpf, ctab, stream);       \
    return g.modulated ? launch_fwd_hp<3, true, T>(g, hd, t, xt, wpf, ctab, stream)           \
                       : launch_fwd_hp<3, false, T>(g, hd, t, xt, wpf, ctab, stream);         \
  } while (0)
  if (dtype == MDCONV_F16) HP_DISPATCH(F16);
  HP_DISPATCH(BF16);
#undef HP_DISPATCH
}

}  // namespace mdconv
