// direct_kernels.hip -- shape-generic VALU kernels (any dtype / groups / kernel size / 2-D+3-D).
//
// These are the correctness-first path and the fallback for shapes the MFMA implicit-GEMM
// kernels do not take (tiny channel counts per group, fp64).  They are fused: no column buffer
// ever exists in HBM (the reference materialises [C*K, step*S_o] columns three times per
// iteration, mdeformable_conv.cu:159, 396-397).
//
//   direct_fwd        : thread = output pixel, TO output channels in registers; one sampling
//                       state per (deformable group, tap) reused by every input channel.
//   direct_bwd_data   : col2im (grad_input scatter) + col2im_coord (grad_offset / grad_mask),
//                       with grad_col = W^T . grad_out recomputed on the fly from an LDS weight
//                       tile; grad_offset / grad_mask are reduced over channels in registers and
//                       flushed once per (dg, tap, pixel) instead of once per sample
//                       (reference: 3 same-address atomics per sample, mdeformable_conv.cu:303-315).
//   direct_bwd_weight : grad_weight / grad_bias, per-thread register tile, wave-shuffle + LDS
//                       block reduction, one atomic per (block, element).
#include "mdconv_common.hpp"
#include "mfma_tile.hpp"   // raw buffer loads

namespace mdconv {

namespace {

constexpr int kThreads = 256;

// The two neighbours along the contiguous axis in ONE load (see make_pairs): 4 bytes for half,
// 8 for float, 16 for double.  Raw buffer loads take any element-aligned byte offset.
template <typename T> struct PairLoad;
template <> struct PairLoad<float> {
  static __device__ __forceinline__ void ld(rsrc_t r, unsigned voff, float &x, float &y) {
    const float2 v = buf_load2(r, (int)voff, 0);
    x = v.x; y = v.y;
  }
};
template <> struct PairLoad<__half> {
  static __device__ __forceinline__ void ld(rsrc_t r, unsigned voff, float &x, float &y) {
    const unsigned bits = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0);
    x = __half2float(__ushort_as_half((unsigned short)(bits & 0xffffu)));
    y = __half2float(__ushort_as_half((unsigned short)(bits >> 16)));
  }
};
template <> struct PairLoad<bf16_t> {
  static __device__ __forceinline__ void ld(rsrc_t r, unsigned voff, float &x, float &y) {
    const unsigned bits = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0);
    x = __uint_as_float(bits << 16);
    y = __uint_as_float(bits & 0xffff0000u);
  }
};
template <> struct PairLoad<double> {
  static __device__ __forceinline__ void ld(rsrc_t r, unsigned voff, double &x, double &y) {
    struct D2 { double x, y; };
    const D2 v = __builtin_bit_cast(D2, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0));
    x = v.x; y = v.y;
  }
};

template <typename T, int ND, bool MOD, typename A>
__device__ __forceinline__ void load_tap(const Geom &g, const T *offset, const T *mask, int b,
                                         int dg, int tap, int pix, const int *o, bool bwd,
                                         TapCoef<ND, A> &tc, A &m) {
  A delta[ND];
  const int64_t obase = ((int64_t)(b * g.DG + dg) * (ND * g.K) + ND * tap) * g.S_o + pix;
#pragma unroll
  for (int a = 0; a < ND; ++a) delta[a] = (A)ld(offset + obase + (int64_t)a * g.S_o);
  int t[ND];
  tap_coords<ND>(g, tap, t);
  make_tap<ND, A>(g, o, t, delta, bwd, tc);
  m = MOD ? (A)ld(mask + ((int64_t)(b * g.DG + dg) * g.K + tap) * g.S_o + pix) : (A)1;
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// PAIR: fetch the corner pairs of the contiguous axis with one load each (2^(ND-1) gathers per
// sample instead of 2^ND: the kernel is bound by gather instructions, cfg3 shape 0.61 -> see
// DESIGN.md); needs >= 2 columns and an input tensor below 4 GiB (32-bit buffer offsets).
template <typename T, int ND, bool MOD, int TO, bool PAIR>
__global__ __launch_bounds__(kThreads) void direct_fwd_kernel(Geom g, int CC, const T *__restrict__ input,
                                                              const T *__restrict__ weight,
                                                              const T *__restrict__ bias,
                                                              const T *__restrict__ offset,
                                                              const T *__restrict__ mask,
                                                              T *__restrict__ output) {
  using A = typename Acc<T>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  A *Ws = reinterpret_cast<A *>(smem_raw);  // [CC][K][TO]
  const int otiles = (g.Og + TO - 1) / TO;
  const int grp = blockIdx.y / otiles;
  const int o0 = (blockIdx.y - grp * otiles) * TO;
  const int n_raw = blockIdx.x * kThreads + threadIdx.x;
  const bool live = n_raw < g.N;
  const int n = live ? n_raw : g.N - 1;
  const int b = n / g.S_o;
  const int pix = n - b * g.S_o;
  int o[ND];
  out_coords<ND>(g, pix, o);

  A acc[TO];
#pragma unroll
  for (int t = 0; t < TO; ++t) acc[t] = (A)0;
  constexpr int NP = 1 << (ND - 1);
  const rsrc_t r_in = make_rsrc(input, (size_t)g.B * g.C * g.S_i * sizeof(T));

  for (int c0 = 0; c0 < g.Cg; c0 += CC) {
    const int cc_n = min(CC, g.Cg - c0);
    __syncthreads();
    for (int i = threadIdx.x; i < cc_n * g.K * TO; i += kThreads) {
      const int t = i % TO;
      const int r = i / TO;  // cc*K + tap
      const int oc = o0 + t;
      Ws[i] = (oc < g.Og) ? (A)ld(weight + ((int64_t)(grp * g.Og + oc) * g.Cg + c0) * g.K + r) : (A)0;
    }
    __syncthreads();
    for (int tap = 0; tap < g.K; ++tap) {
      int cur_dg = -1;
      TapCoef<ND, A> tc;
      A m = (A)1;
      int pidx[NP];
      A px[NP], py[NP];
      bool prx[NP], pry[NP];
      for (int cc = 0; cc < cc_n; ++cc) {
        const int c = grp * g.Cg + c0 + cc;
        const int dg = c / g.Cdg;
        if (dg != cur_dg) {
          load_tap<T, ND, MOD, A>(g, offset, mask, b, dg, tap, pix, o, false, tc, m);
          if (PAIR) {
            make_pairs<ND, A>(g, tc, m, pidx, px, py);   // mask folded into the weights
            make_pairs_read<ND, A>(g, tc, prx, pry);
          }
          cur_dg = dg;
        }
        A val = (A)0;
        if (PAIR) {
          const unsigned plane_off = (unsigned)(b * g.C + c) * (unsigned)g.S_i;
#pragma unroll
          for (int pi = 0; pi < NP; ++pi) {
            A x, y;
            PairLoad<T>::ld(r_in, (plane_off + (unsigned)pidx[pi]) * (unsigned)sizeof(T), x, y);
            // an element the reference never reads (weight 0) must not turn a non-finite neighbour into NaN
            val += (prx[pi] ? px[pi] * x : (A)0) + (pry[pi] ? py[pi] * y : (A)0);
          }
        } else {
          const T *plane = input + (int64_t)(b * g.C + c) * g.S_i;
#pragma unroll
          for (int ci = 0; ci < (1 << ND); ++ci)
            if (corner_is_read<ND, A>(tc, ci))
              val += corner_weight<ND, A>(tc, ci) * (A)ld(plane + corner_index<ND, A>(tc, ci));
          val *= m;
        }
        const A *wrow = Ws + (cc * g.K + tap) * TO;
#pragma unroll
        for (int t = 0; t < TO; ++t) acc[t] += wrow[t] * val;
      }
    }
  }
  if (live) {
#pragma unroll
    for (int t = 0; t < TO; ++t) {
      const int oc = o0 + t;
      if (oc < g.Og) {
        const int och = grp * g.Og + oc;
        const A bv = g.with_bias ? (A)ld(bias + och) : (A)0;
        st(output + (int64_t)(b * g.O + och) * g.S_o + pix, acc[t] + bv);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward: grad_input / grad_offset / grad_mask
// ------------------------------------------------------------------------------------------
template <typename T, int ND, bool MOD, int TO>
__global__ __launch_bounds__(kThreads) void direct_bwd_data_kernel(
    Geom g, int CC, const T *__restrict__ input, const T *__restrict__ weight,
    const T *__restrict__ offset, const T *__restrict__ mask, const T *__restrict__ grad_output,
    T *__restrict__ grad_input, T *__restrict__ grad_offset, T *__restrict__ grad_mask) {
  using A = typename Acc<T>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  A *Ws = reinterpret_cast<A *>(smem_raw);  // [CC][K][TO]
  const int otiles = (g.Og + TO - 1) / TO;
  const int grp = blockIdx.y / otiles;
  const int o0 = (blockIdx.y - grp * otiles) * TO;
  const int n_raw = blockIdx.x * kThreads + threadIdx.x;
  const bool live = n_raw < g.N;
  const int n = live ? n_raw : g.N - 1;
  const int b = n / g.S_o;
  const int pix = n - b * g.S_o;
  int o[ND];
  out_coords<ND>(g, pix, o);

  A go[TO];
#pragma unroll
  for (int t = 0; t < TO; ++t) {
    const int oc = o0 + t;
    go[t] = (live && oc < g.Og)
                ? (A)ld(grad_output + (int64_t)(b * g.O + grp * g.Og + oc) * g.S_o + pix)
                : (A)0;
  }

  for (int c0 = 0; c0 < g.Cg; c0 += CC) {
    const int cc_n = min(CC, g.Cg - c0);
    __syncthreads();
    for (int i = threadIdx.x; i < cc_n * g.K * TO; i += kThreads) {
      const int t = i % TO;
      const int r = i / TO;
      const int oc = o0 + t;
      Ws[i] = (oc < g.Og) ? (A)ld(weight + ((int64_t)(grp * g.Og + oc) * g.Cg + c0) * g.K + r) : (A)0;
    }
    __syncthreads();
    if (!live) continue;
    for (int tap = 0; tap < g.K; ++tap) {
      int cur_dg = -1;
      TapCoef<ND, A> tc;
      A m = (A)1;
      A goff[ND];
      A gm = (A)0;
      auto flush = [&]() {
        if (cur_dg < 0) return;
        if (!g.range_gate || tc.inside) {
          const int64_t obase = ((int64_t)(b * g.DG + cur_dg) * (ND * g.K) + ND * tap) * g.S_o + pix;
#pragma unroll
          for (int a = 0; a < ND; ++a)
            if (goff[a] != (A)0) atomic_add(grad_offset + obase + (int64_t)a * g.S_o, goff[a] * m);
        }
        if (MOD && gm != (A)0)
          atomic_add(grad_mask + ((int64_t)(b * g.DG + cur_dg) * g.K + tap) * g.S_o + pix, gm);
      };
      for (int cc = 0; cc < cc_n; ++cc) {
        const int c = grp * g.Cg + c0 + cc;
        const int dg = c / g.Cdg;
        if (dg != cur_dg) {
          flush();
          load_tap<T, ND, MOD, A>(g, offset, mask, b, dg, tap, pix, o, true, tc, m);
          cur_dg = dg;
#pragma unroll
          for (int a = 0; a < ND; ++a) goff[a] = (A)0;
          gm = (A)0;
        }
        // grad_col(c, tap, n) restricted to this block's TO output channels (GEMM-1 of the
        // reference, mdeformable_conv.cu:417-419); every later use is linear in it.
        const A *wrow = Ws + (cc * g.K + tap) * TO;
        A gcol = (A)0;
#pragma unroll
        for (int t = 0; t < TO; ++t) gcol += wrow[t] * go[t];
        const int64_t pbase = (int64_t)(b * g.C + c) * g.S_i;
        A v[1 << ND];
        A val = (A)0;
#pragma unroll
        for (int ci = 0; ci < (1 << ND); ++ci) {
          const int idx = corner_index<ND, A>(tc, ci);
          v[ci] = corner_is_read<ND, A>(tc, ci) ? (A)ld(input + pbase + idx) : (A)0;   // never read by the reference otherwise
          val += corner_weight<ND, A>(tc, ci) * v[ci];
          const A wa = corner_weight_atom<ND, A>(tc, ci) * m * gcol;  // w * dval, :282-293
          if (wa != (A)0) atomic_add(grad_input + pbase + idx, wa);
        }
#pragma unroll
        for (int a = 0; a < ND; ++a) {
          A dv = (A)0;
#pragma unroll
          for (int ci = 0; ci < (1 << ND); ++ci) dv += corner_dweight<ND, A>(tc, ci, a) * v[ci];
          goff[a] += dv * gcol;
        }
        gm += val * gcol;
      }
      flush();
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward: grad_weight / grad_bias
// ------------------------------------------------------------------------------------------
template <typename A> __device__ __forceinline__ A wave_sum(A v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <typename T, int ND, bool MOD, int TO, int CC>
__global__ __launch_bounds__(kThreads) void direct_bwd_weight_kernel(
    Geom g, int n_per_split, const T *__restrict__ input, const T *__restrict__ offset,
    const T *__restrict__ mask, const T *__restrict__ grad_output, T *__restrict__ grad_weight,
    T *__restrict__ grad_bias) {
  using A = typename Acc<T>::type;
  __shared__ A red[kThreads / 64][CC * TO + TO];
  const int otiles = (g.Og + TO - 1) / TO;
  const int cchunks = (g.Cg + CC - 1) / CC;
  // blockIdx.x = ((grp * otiles + otile) * cchunks + cchunk) * K + tap
  int id = blockIdx.x;
  const int tap = id % g.K;
  id /= g.K;
  const int cchunk = id % cchunks;
  id /= cchunks;
  const int otile = id % otiles;
  const int grp = id / otiles;
  const int o0 = otile * TO;
  const int c0 = cchunk * CC;
  const int cc_n = min(CC, g.Cg - c0);
  const bool do_bias = g.with_bias && tap == 0 && cchunk == 0;

  A acc[CC][TO];
  A bsum[TO];
#pragma unroll
  for (int cc = 0; cc < CC; ++cc)
#pragma unroll
    for (int t = 0; t < TO; ++t) acc[cc][t] = (A)0;
#pragma unroll
  for (int t = 0; t < TO; ++t) bsum[t] = (A)0;

  const int n_begin = blockIdx.y * n_per_split;
  const int n_end = min(g.N, n_begin + n_per_split);
  for (int n = n_begin + threadIdx.x; n < n_end; n += kThreads) {
    const int b = n / g.S_o;
    const int pix = n - b * g.S_o;
    int o[ND];
    out_coords<ND>(g, pix, o);
    A go[TO];
#pragma unroll
    for (int t = 0; t < TO; ++t) {
      const int oc = o0 + t;
      go[t] = (oc < g.Og) ? (A)ld(grad_output + (int64_t)(b * g.O + grp * g.Og + oc) * g.S_o + pix)
                          : (A)0;
      bsum[t] += go[t];
    }
    int cur_dg = -1;
    TapCoef<ND, A> tc;
    A m = (A)1;
#pragma unroll
    for (int cc = 0; cc < CC; ++cc) {
      if (cc < cc_n) {
        const int c = grp * g.Cg + c0 + cc;
        const int dg = c / g.Cdg;
        if (dg != cur_dg) {
          load_tap<T, ND, MOD, A>(g, offset, mask, b, dg, tap, pix, o, true, tc, m);
          cur_dg = dg;
        }
        const T *plane = input + (int64_t)(b * g.C + c) * g.S_i;
        A val = (A)0;
#pragma unroll
        for (int ci = 0; ci < (1 << ND); ++ci)
          if (corner_is_read<ND, A>(tc, ci))
            val += corner_weight<ND, A>(tc, ci) * (A)ld(plane + corner_index<ND, A>(tc, ci));
        val *= m;  // the re-materialised forward column, mdeformable_conv.cu:316
#pragma unroll
        for (int t = 0; t < TO; ++t) acc[cc][t] += go[t] * val;
      }
    }
  }
  // block reduction: wave shuffles, then LDS across the 4 waves
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int cc = 0; cc < CC; ++cc)
#pragma unroll
    for (int t = 0; t < TO; ++t) {
      const A s = wave_sum(acc[cc][t]);
      if (lane == 0) red[wave][cc * TO + t] = s;
    }
#pragma unroll
  for (int t = 0; t < TO; ++t) {
    const A s = wave_sum(bsum[t]);
    if (lane == 0) red[wave][CC * TO + t] = s;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < CC * TO + TO; i += kThreads) {
    A s = (A)0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) s += red[w][i];
    if (i < CC * TO) {
      const int cc = i / TO, t = i - cc * TO;
      const int oc = o0 + t;
      if (cc < cc_n && oc < g.Og && s != (A)0)
        atomic_add(grad_weight + ((int64_t)(grp * g.Og + oc) * g.Cg + c0 + cc) * g.K + tap, s);
    } else if (do_bias) {
      const int t = i - CC * TO;
      const int oc = o0 + t;
      if (oc < g.Og && s != (A)0) atomic_add(grad_bias + grp * g.Og + oc, s);
    }
  }
}

// dynamic LDS above the 64 KB default needs an explicit opt-in per kernel (gfx950: 160 KB per workgroup)
constexpr size_t kMaxLds = 160 * 1024;
template <typename K> int allow_lds(K kernel, size_t smem) {
  if (smem <= 64 * 1024) return MDCONV_OK;
  hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return MDCONV_ELAUNCH; }
  return MDCONV_OK;
}

int pick_cc(const Geom &g, int TO, size_t elem) {
  const int budget = 32 * 1024;
  int cc = (int)(budget / ((size_t)g.K * TO * elem));
  if (cc < 1) cc = 1;
  if (cc > g.Cg) cc = g.Cg;
  return cc;
}

template <typename T, int ND, bool MOD>
int launch_fwd(const Geom &g, const Tensors &t, hipStream_t stream) {
  using A = typename Acc<T>::type;
  constexpr int TO = 16;
  const int cc = pick_cc(g, TO, sizeof(A));
  const size_t smem = (size_t)cc * g.K * TO * sizeof(A);
  const int otiles = (g.Og + TO - 1) / TO;
  if (smem > kMaxLds || (int64_t)g.G * otiles > 65535) {
    set_error("direct_fwd: kernel volume K=%d / group count beyond the LDS weight tile or the grid", g.K);
    return MDCONV_EUNSUPPORTED;
  }
  dim3 grid((g.N + kThreads - 1) / kThreads, g.G * otiles);
  int rc_lds;
  const bool pair = g.in_sz[g.nd - 1] >= 2 &&
                    (size_t)g.B * g.C * g.S_i * sizeof(T) < 0xfffffff0ull;
  if ((rc_lds = pair ? allow_lds(direct_fwd_kernel<T, ND, MOD, TO, true>, smem)
                     : allow_lds(direct_fwd_kernel<T, ND, MOD, TO, false>, smem)))
    return rc_lds;
  if (pair)
    hipLaunchKernelGGL((direct_fwd_kernel<T, ND, MOD, TO, true>), grid, dim3(kThreads), smem, stream, g,
                       cc, (const T *)t.input, (const T *)t.weight, (const T *)t.bias,
                       (const T *)t.offset, (const T *)t.mask, (T *)t.output);
  else
    hipLaunchKernelGGL((direct_fwd_kernel<T, ND, MOD, TO, false>), grid, dim3(kThreads), smem, stream, g,
                       cc, (const T *)t.input, (const T *)t.weight, (const T *)t.bias,
                       (const T *)t.offset, (const T *)t.mask, (T *)t.output);
  return check_launch("direct_fwd");
}

template <typename T, int ND, bool MOD>
int launch_bwd(const Geom &g, const Tensors &t, hipStream_t stream, int parts) {
  using A = typename Acc<T>::type;
  if (parts & 1) {
    constexpr int TO = 32;
    const int cc = pick_cc(g, TO, sizeof(A));
    const size_t smem = (size_t)cc * g.K * TO * sizeof(A);
    const int otiles = (g.Og + TO - 1) / TO;
    if (smem > kMaxLds || (int64_t)g.G * otiles > 65535) {
      set_error("direct_bwd_data: kernel volume K=%d / group count beyond the LDS weight tile or the grid", g.K);
      return MDCONV_EUNSUPPORTED;
    }
    int rc_lds = allow_lds(direct_bwd_data_kernel<T, ND, MOD, TO>, smem);
    if (rc_lds) return rc_lds;
    dim3 grid((g.N + kThreads - 1) / kThreads, g.G * otiles);
    hipLaunchKernelGGL((direct_bwd_data_kernel<T, ND, MOD, TO>), grid, dim3(kThreads), smem, stream,
                       g, cc, (const T *)t.input, (const T *)t.weight, (const T *)t.offset,
                       (const T *)t.mask, (const T *)t.grad_output, (T *)t.grad_input,
                       (T *)t.grad_offset, (T *)t.grad_mask);
    int rc = check_launch("direct_bwd_data");
    if (rc) return rc;
  }
  if (parts & 2) {
    constexpr int TO = 8, CC = 8;
    const int otiles = (g.Og + TO - 1) / TO;
    const int cchunks = (g.Cg + CC - 1) / CC;
    const int64_t gx = (int64_t)g.G * otiles * cchunks * g.K;
    if (gx > 0x7fffffffLL) {
      set_error("direct_bwd_weight: grid too large");
      return MDCONV_EUNSUPPORTED;
    }
    // split the pixel range until ~2048 blocks are in flight, >= 8 iterations per thread
    int splits = (int)((2048 + gx - 1) / gx);
    const int max_splits = (g.N + kThreads * 8 - 1) / (kThreads * 8);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    const int n_per_split = (g.N + splits - 1) / splits;
    dim3 grid((unsigned)gx, (unsigned)((g.N + n_per_split - 1) / n_per_split));
    hipLaunchKernelGGL((direct_bwd_weight_kernel<T, ND, MOD, TO, CC>), grid, dim3(kThreads), 0,
                       stream, g, n_per_split, (const T *)t.input, (const T *)t.offset,
                       (const T *)t.mask, (const T *)t.grad_output, (T *)t.grad_weight,
                       (T *)t.grad_bias);
    return check_launch("direct_bwd_weight");
  }
  return MDCONV_OK;
}

template <typename T> int dispatch_fwd(const Geom &g, const Tensors &t, hipStream_t s) {
  if (g.nd == 2) return g.modulated ? launch_fwd<T, 2, true>(g, t, s) : launch_fwd<T, 2, false>(g, t, s);
  return g.modulated ? launch_fwd<T, 3, true>(g, t, s) : launch_fwd<T, 3, false>(g, t, s);
}
template <typename T> int dispatch_bwd(const Geom &g, const Tensors &t, hipStream_t s, int parts) {
  if (g.nd == 2) return g.modulated ? launch_bwd<T, 2, true>(g, t, s, parts) : launch_bwd<T, 2, false>(g, t, s, parts);
  return g.modulated ? launch_bwd<T, 3, true>(g, t, s, parts) : launch_bwd<T, 3, false>(g, t, s, parts);
}

}  // namespace

int direct_forward(const Geom &g, int dtype, const Tensors &t, hipStream_t stream) {
  switch (dtype) {
    case MDCONV_F32: return dispatch_fwd<float>(g, t, stream);
    case MDCONV_F16: return dispatch_fwd<__half>(g, t, stream);
    case MDCONV_F64: return dispatch_fwd<double>(g, t, stream);
    case MDCONV_BF16: return dispatch_fwd<bf16_t>(g, t, stream);
  }
  set_error("unknown dtype %d", dtype);
  return MDCONV_EINVAL;
}

int direct_backward(const Geom &g, int dtype, const Tensors &t, hipStream_t stream, int parts) {
  switch (dtype) {
    case MDCONV_F32: return dispatch_bwd<float>(g, t, stream, parts);
    case MDCONV_F16: return dispatch_bwd<__half>(g, t, stream, parts);
    case MDCONV_F64: return dispatch_bwd<double>(g, t, stream, parts);
    case MDCONV_BF16: return dispatch_bwd<bf16_t>(g, t, stream, parts);
  }
  set_error("unknown dtype %d", dtype);
  return MDCONV_EINVAL;
}

}  // namespace mdconv
