// hp_bwd.hip -- the fused backward kernel of the native 16-bit path (fp16 / bf16, gfx950).
//
// Reference structure (mdeformable_conv.cu:412-444; 3-D: mdeformable_conv3d.cu:515-560): GEMM-1
// grad_col = W^T . grad_out, the per-sample gradient kernel (recompute the sample, atomics into
// grad_input / grad_offset / grad_mask, rewrite `columns`), GEMM-2 grad_W += grad_out . columns^T.
// At the fp16 matrix rate all of that is bound by the corner gathers, so this kernel gathers every
// corner ONCE and feeds all consumers from registers:
//
//   workgroup = (tap, range of 32-pixel tiles); wave w = input channels [32w, 32w+32)   ("tap
//   stationary": W^T[tap] lives in registers and the grad_W[tap] accumulators persist over the
//   whole pixel range -- split-K over ranges, reduced by hp_reduce_gw_kernel)
//   per tile:
//     grad_out tile -> LDS in both orientations ([pixel][o] for GEMM-1, [o][pixel] for GEMM-2)
//     GEMM-1   gc[c, n]  = sum_o W[o, c, tap] grad_out[o, n]      M = channels (rows permuted so a
//              lane owns 16 CONSECUTIVE channels of its pixel), N = pixels, K = o
//     gather   2^ND corners x 16 channels of the lane's pixel (2 x 16-byte loads per corner)
//     S[ci]    = sum_c gc[c] x[ci][c]   ->  grad_mask = sum_ci w[ci] S[ci],
//              grad_offset_a = mask * sum_ci dw_a[ci] S[ci]   (reduced over the half-waves by a
//              shuffle, over the waves of a deformable group through LDS; single owner, no atomics)
//     grad_col row (raw gc, 16-bit) -> workspace [b][tap][pix][c] for the col2im gather
//     col[c]   = mask * sum_ci w[ci] x[ci][c]  -> LDS, transposed  -> B operand of
//     GEMM-2   grad_W[o, c] += sum_n grad_out[o, n] col[n, c]      M = o, N = channels, K = pixels
//   The first wave of every deformable group also counts the scatter targets of its samples (the
//   first pass of the CSR build, hp_col2im.hip) with fire-and-forget integer atomics.
#include "hp_kernels.hpp"

namespace mdconv {

namespace {


template <int ND, bool MOD, typename T, int WAVES, int NKS>
__global__ __launch_bounds__(64 * WAVES, (WAVES >= 8 || NKS >= 16 || ND == 3) ? 1 : 2) void hp_bwd_kernel(
    Geom g, HpDims hd, const typename T::Raw *__restrict__ xt, const U4 *__restrict__ wpb,
    const int4 *__restrict__ btab, const typename T::Raw *__restrict__ gout,
    const typename T::Raw *__restrict__ offset, const typename T::Raw *__restrict__ mask,
    typename T::Raw *__restrict__ gcol, typename T::Raw *__restrict__ grad_offset,
    typename T::Raw *__restrict__ grad_mask, float *__restrict__ part, int *__restrict__ cnt) {
  using Raw = typename T::Raw;
  constexpr int NC = 1 << ND, NP = NC / 2;
  constexpr int MB2 = NKS / 2;
  constexpr int NT = 64 * WAVES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int OpL = hd.OpL;
  const int pitch_po = OpL + 8;                 // [pixel][o] rows
  Raw *Gop = reinterpret_cast<Raw *>(smem);     // [OpL][kPP]
  Raw *Gpo = Gop + OpL * kPP;                   // [32][pitch_po]
  Raw *colT = Gpo + 32 * pitch_po;              // [WAVES][32][kPP]
  float *red = reinterpret_cast<float *>(colT + WAVES * 32 * kPP);   // [WAVES][ND + 1][32]

  const int tid = threadIdx.x, lane = tid & 63, kh = lane >> 5, pl = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cblk = wave;
  const bool active = cblk < hd.cblks;
  const int tap = blockIdx.y, range = blockIdx.x;
  const int t_lo = range * hd.tiles_per_range;
  const int t_hi = min(t_lo + hd.tiles_per_range, hd.ntiles);
  const int o_base = active ? btab[cblk].x : 0;
  const int wpd = g.DG == 1 ? WAVES : g.Cdg / 32;   // waves per deformable group
  const int dg = g.DG == 1 ? 0 : min(cblk * 32, g.C - 1) / g.Cdg;
  const bool counter = active && (g.DG == 1 ? cblk == 0 : cblk % wpd == 0);

  const rsrc_t r_xt = make_rsrc(xt, (size_t)g.B * g.S_i * hd.Cp * 2);

  // W^T[tap] fragments of this wave's channel block: resident for the whole pixel range
  U4 wf[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks)
    wf[ks] = active ? wpb[(((int64_t)tap * hd.cblks + cblk) * NKS + ks) * 64 + lane] : U4{0, 0, 0, 0};

  f32x16 acc2[MB2];
#pragma unroll
  for (int i = 0; i < MB2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[i][r] = 0.f;

  int tcd[ND];
  tap_coords<ND>(g, tap, tcd);

  // ---- grad_out tile loader: item = (o, pixel octet) ----
  const bool vec_ok = (g.S_o & 7) == 0;
  auto load_item = [&](int item, int n0) -> U4 {
    const int o = item >> 2, oct = item & 3;
    const int nn = n0 + oct * 8;
    U4 v = {0, 0, 0, 0};
    if (o < g.O && nn < g.N) {
      if (vec_ok) {
        const int bb = nn / g.S_o, pp = nn - bb * g.S_o;
        v = *reinterpret_cast<const U4 *>(gout + ((int64_t)bb * g.O + o) * g.S_o + pp);
      } else {
        unsigned short e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int n2 = nn + j;
          const int bb = min(n2, g.N - 1) / g.S_o, pp = min(n2, g.N - 1) - bb * g.S_o;
          const Raw x = gout[((int64_t)bb * g.O + o) * g.S_o + pp];
          e[j] = n2 < g.N ? __builtin_bit_cast(unsigned short, x) : (unsigned short)0;
        }
        v.x = e[0] | ((u32)e[1] << 16); v.y = e[2] | ((u32)e[3] << 16);
        v.z = e[4] | ((u32)e[5] << 16); v.w = e[6] | ((u32)e[7] << 16);
      }
    }
    return v;
  };
  auto store_item = [&](int item, const U4 &v) {
    const int o = item >> 2, oct = item & 3;
    *reinterpret_cast<U4 *>(Gop + o * kPP + oct * 8) = v;
    unsigned short *po = reinterpret_cast<unsigned short *>(Gpo) + (oct * 8) * pitch_po + o;
    po[0 * pitch_po] = (unsigned short)(v.x & 0xffffu); po[1 * pitch_po] = (unsigned short)(v.x >> 16);
    po[2 * pitch_po] = (unsigned short)(v.y & 0xffffu); po[3 * pitch_po] = (unsigned short)(v.y >> 16);
    po[4 * pitch_po] = (unsigned short)(v.z & 0xffffu); po[5 * pitch_po] = (unsigned short)(v.z >> 16);
    po[6 * pitch_po] = (unsigned short)(v.w & 0xffffu); po[7 * pitch_po] = (unsigned short)(v.w >> 16);
  };
  const int nitems = OpL * 4;

  // ---- flush of the previous tile's coordinate gradients (single owner per element) ----
  auto flush = [&](int n0p) {
    const int items = g.DG * (ND + 1) * 32;
    for (int x = tid; x < items; x += NT) {
      const int p = x & 31, a = (x >> 5) % (ND + 1), dgi = x / (32 * (ND + 1));
      const int nn = n0p + p;
      if (nn < g.N && (MOD || a < ND)) {
        float sum = 0.f;
        const int w0 = dgi * wpd, w1 = min(w0 + wpd, hd.cblks);
        for (int w = w0; w < w1; ++w) sum += red[(w * (ND + 1) + a) * 32 + p];
        const int bb = nn / g.S_o, pp = nn - bb * g.S_o;
        const int64_t seg = (int64_t)bb * g.DG + dgi;
        Raw *dst = a < ND ? grad_offset + (seg * (ND * g.K) + ND * tap + a) * g.S_o + pp
                          : grad_mask + (seg * g.K + tap) * g.S_o + pp;
        T::stf(dst, g.acc_data ? T::ldf(dst) + sum : sum);
      }
    }
  };

  // offsets / mask of the lane's pixel, one tile ahead
  Raw dlr[ND], mlr;   // raw 16-bit values until they are needed (a conversion here is a use of the load where it is issued)
  auto fetch = [&](int tile) {
    const int nn = min(tile * 32 + pl, g.N - 1);
    const int bb = nn / g.S_o, pp = nn - bb * g.S_o;
    const int64_t seg = (int64_t)bb * g.DG + dg;
    const int64_t ob = (seg * (ND * g.K) + ND * tap) * g.S_o + pp;
#pragma unroll
    for (int a = 0; a < ND; ++a) dlr[a] = offset[ob + (int64_t)a * g.S_o];
    if (MOD) mlr = mask[(seg * g.K + tap) * g.S_o + pp];
  };
  if (t_lo < t_hi) fetch(t_lo);

  for (int tile = t_lo; tile < t_hi; ++tile) {
    const int n0 = tile * 32;
    const int n_raw = n0 + pl;
    const bool live = n_raw < g.N;
    const int n = live ? n_raw : g.N - 1;
    const int b = n / g.S_o, pix = n - b * g.S_o;

    // grad_out tile: first two items per thread in flight across the barrier
    U4 gi0 = {0, 0, 0, 0}, gi1 = {0, 0, 0, 0};
    if (tid < nitems) gi0 = load_item(tid, n0);
    if (tid + NT < nitems) gi1 = load_item(tid + NT, n0);

    // ---- sampling state (backward gating flavours) and the corner gathers ----
    int oc[ND];
    out_coords<ND>(g, pix, oc);
    TapCoef<ND, float> tc;
    float dl[ND];
#pragma unroll
    for (int a = 0; a < ND; ++a) dl[a] = T::ldf(&dlr[a]);
    make_tap<ND, float>(g, oc, tcd, dl, true, tc);
    const float m_n = MOD ? T::ldf(&mlr) : 1.f;
    if (tile + 1 < t_hi) fetch(tile + 1);
    HpCorners<ND> hc;
    hp_corners<ND>(tc, hc);
    const float mg = (!g.range_gate || tc.inside) ? m_n : 0.f;
    U4 x0[NC], x1[NC];
    {
      const int cb2 = (cblk * 32 + 16 * kh) * 2;
#pragma unroll
      for (int ci = 0; ci < NC; ++ci) {
        const int vo = (active && hc.idx[ci] >= 0) ? ((b * g.S_i + hc.idx[ci]) * hd.Cp) * 2 + cb2 : kHpOob;
        x0[ci] = buf_load4u(r_xt, vo, 0);
        x1[ci] = buf_load4u(r_xt, vo + 16, 0);
      }
    }
    if (counter && kh == 0 && live) {
      // scatter anchor of this sample (first pass of the CSR build, hp_col2im.hip)
      SampleAnchor<ND> sa;
      sample_anchor<ND>(g, tc, 1.f, sa);
      if (sa.on) atomicAdd(cnt + ((int64_t)b * g.DG + dg) * hp_anchor_space(g) + sa.qa, 1);
    }

    __syncthreads();   // every wave is done with the previous tile's LDS tiles and partial sums
    if (tile > t_lo) flush(n0 - 32);
    if (tid < nitems) store_item(tid, gi0);
    if (tid + NT < nitems) store_item(tid + NT, gi1);
    for (int item = tid + 2 * NT; item < nitems; item += NT) store_item(item, load_item(item, n0));
    __syncthreads();

    if (active) {
      // ---- GEMM-1 ----
      f32x16 gc;
#pragma unroll
      for (int r = 0; r < 16; ++r) gc[r] = 0.f;
      const Raw *bp = Gpo + pl * pitch_po + o_base + 8 * kh;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
        gc = T::mfma(wf[ks], *reinterpret_cast<const U4 *>(bp + ks * 16), gc);
      float gcv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) gcv[r] = gc[r];

      // grad_col in the storage type: the row written for col2im and the operand of the corner sums
      float(&g0)[8] = *reinterpret_cast<float(*)[8]>(gcv);
      float(&g1)[8] = *reinterpret_cast<float(*)[8]>(gcv + 8);
      const U4 gp0 = pack8<T>(g0), gp1 = pack8<T>(g1);

      // ---- corner sums and the forward column values ----
      float col[16], S[NC];
#pragma unroll
      for (int r = 0; r < 16; ++r) col[r] = 0.f;
#pragma unroll
      for (int ci = 0; ci < NC; ++ci) {
        S[ci] = dot8<T>(dot8<T>(0.f, x0[ci], gp0), x1[ci], gp1);
        const float wm = hc.w[ci] * m_n;
        float(&c0)[8] = *reinterpret_cast<float(*)[8]>(col);
        float(&c1)[8] = *reinterpret_cast<float(*)[8]>(col + 8);
        mac8<T>(c0, x0[ci], wm);
        mac8<T>(c1, x1[ci], wm);
      }

      // ---- grad_col row ----
      if (live) {
        Raw *row = gcol + (((int64_t)b * g.K + tap) * g.S_o + pix) * hd.Cp + cblk * 32 + 16 * kh;
        *reinterpret_cast<U4 *>(row) = gp0;
        *reinterpret_cast<U4 *>(row + 8) = gp1;
      }

      // ---- coordinate gradients of this wave's channels ----
      {
        float gm = 0.f, goff[ND];
#pragma unroll
        for (int ci = 0; ci < NC; ++ci) gm = fmaf(hc.w[ci], S[ci], gm);
#pragma unroll
        for (int a = 0; a < ND; ++a) {
          goff[a] = 0.f;
#pragma unroll
          for (int ci = 0; ci < NC; ++ci) goff[a] = fmaf(corner_dweight<ND, float>(tc, ci, a), S[ci], goff[a]);
          goff[a] *= mg;
          goff[a] += __shfl_xor(goff[a], 32, 64);
        }
        gm += __shfl_xor(gm, 32, 64);
        if (kh == 0) {
          float *rp = red + (wave * (ND + 1)) * 32 + pl;
#pragma unroll
          for (int a = 0; a < ND; ++a) rp[a * 32] = goff[a];
          rp[ND * 32] = gm;
        }
      }

      // ---- GEMM-2: col -> LDS (transposed) -> B fragments ----
      Raw *ct = colT + wave * 32 * kPP;
#pragma unroll
      for (int r = 0; r < 16; ++r) T::stf(ct + (16 * kh + r) * kPP + pl, col[r]);
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        const U4 bc = *reinterpret_cast<const U4 *>(ct + pl * kPP + ks2 * 16 + 8 * kh);
#pragma unroll
        for (int ob = 0; ob < MB2; ++ob) {
          const U4 a = *reinterpret_cast<const U4 *>(Gop + (o_base + ob * 32 + pl) * kPP + ks2 * 16 + 8 * kh);
          acc2[ob] = T::mfma(a, bc, acc2[ob]);
        }
      }
    }
  }
  if (t_lo < t_hi) {
    __syncthreads();
    flush((t_hi - 1) * 32);
  }
  if (active) {
    float4 *dst = reinterpret_cast<float4 *>(
        part + ((((int64_t)tap * hd.ranges + range) * hd.cblks + cblk) * MB2) * 1024 + lane * 16);
#pragma unroll
    for (int ob = 0; ob < MB2; ++ob)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        dst[ob * 256 + q] = make_float4(acc2[ob][4 * q], acc2[ob][4 * q + 1], acc2[ob][4 * q + 2], acc2[ob][4 * q + 3]);
  }
}

}  // namespace

size_t hp_bwd_lds_bytes(const Geom &g, const HpDims &hd) {
  return (size_t)hd.OpL * kPP * 2 + (size_t)32 * (hd.OpL + 8) * 2 + (size_t)hd.waves * 32 * kPP * 2 +
         (size_t)hd.waves * (g.nd + 1) * 32 * 4;
}

template <int ND, bool MOD, typename T, int WAVES, int NKS>
static int launch_bwd_hp(const Geom &g, const HpDims &hd, const Tensors &t, const void *xt,
                         const void *wpb, const int4 *btab, void *gcol, float *part, int *cnt,
                         hipStream_t stream) {
  using Raw = typename T::Raw;
  const size_t lds = hp_bwd_lds_bytes(g, hd);
  if (lds > 64 * 1024) {
    hipError_t ea = hipFuncSetAttribute((const void *)hp_bwd_kernel<ND, MOD, T, WAVES, NKS>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ea != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(ea)); return MDCONV_ELAUNCH; }
  }
  hipLaunchKernelGGL((hp_bwd_kernel<ND, MOD, T, WAVES, NKS>), dim3(hd.ranges, g.K), dim3(64 * WAVES), lds,
                     stream, g, hd, (const Raw *)xt, (const U4 *)wpb, btab, (const Raw *)t.grad_output,
                     (const Raw *)t.offset, (const Raw *)t.mask, (Raw *)gcol, (Raw *)t.grad_offset,
                     (Raw *)t.grad_mask, part, cnt);
  return check_launch("hp_bwd");
}

template <int ND, bool MOD, typename T>
static int dispatch_bwd_hp(const Geom &g, const HpDims &hd, const Tensors &t, const void *xt,
                           const void *wpb, const int4 *btab, void *gcol, float *part, int *cnt,
                           hipStream_t stream) {
#define HP_BWD(W, N) return launch_bwd_hp<ND, MOD, T, W, N>(g, hd, t, xt, wpb, btab, gcol, part, cnt, stream)
#define HP_BWD_W(W)                                                                            \
  switch (hd.nks) {                                                                            \
    case 2: HP_BWD(W, 2);                                                                      \
    case 4: HP_BWD(W, 4);                                                                      \
    case 8: HP_BWD(W, 8);                                                                      \
    default: HP_BWD(W, 16);                                                                    \
  }
  switch (hd.waves) {
    case 1: HP_BWD_W(1);
    case 2: HP_BWD_W(2);
    case 4: HP_BWD_W(4);
    default: HP_BWD_W(8);
  }
#undef HP_BWD_W
#undef HP_BWD
}

int hp_backward_launch(const Geom &g, const HpDims &hd, int dtype, const Tensors &t, const void *xt,
                       const void *wpb, const int4 *btab, void *gcol, float *part, int *cnt,
                       hipStream_t stream) {
#define HP_DISPATCH(T)                                                                            \
  do {                                                                                            \
    if (g.nd == 2)                                                                                \
      return g.modulated ? dispatch_bwd_hp<2, true, T>(g, hd, t, xt, wpb, btab, gcol, part, cnt, stream)  \
                         : dispatch_bwd_hp<2, false, T>(g, hd, t, xt, wpb, btab, gcol, part, cnt, stream); \
    return g.modulated ? dispatch_bwd_hp<3, true, T>(g, hd, t, xt, wpb, btab, gcol, part, cnt, stream)    \
                       : dispatch_bwd_hp<3, false, T>(g, hd, t, xt, wpb, btab, gcol, part, cnt, stream);   \
  } while (0)
  if (dtype == MDCONV_F16) HP_DISPATCH(F16);
  HP_DISPATCH(BF16);
#undef HP_DISPATCH
}

}  // namespace mdconv
