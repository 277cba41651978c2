// mdconv_common.hpp -- shared host/device definitions for the gfx950 deformable-conv kernels.
//
// Sampling semantics restated from the reference:
//   p_a   = o_a*stride_a - pad_a + tap_a*dil_a + delta_a            (mdeformable_conv.cu:78-79)
//   low_a = floor(p_a), d_a = p_a - low_a; a corner contributes iff it lies inside the image
//   (mdeformable_conv.cu:9-34, 256-267); validity is separable per axis, so it is folded into
//   per-axis weights (an invalid side gets weight 0 and a clamped, always-in-bounds index).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/mdconv.h"

namespace mdconv {

#define MDCONV_EPS 1.192092896e-07F  // reference src/config.h:18

// Device-side geometry (POD, passed by value as a kernel argument).
struct Geom {
  int nd, B, C, O, G, DG, K;
  int in_sz[3], out_sz[3], ksz[3], stride[3], pad[3], dil[3];
  int S_i, S_o;     // spatial volumes (input / output), < 2^31
  int Cg, Og, Cdg;  // channels per conv group (in / out), channels per deformable group
  int N;            // B * S_o, flattened output-pixel count, < 2^31
  // backward gating flavours of the four reference files
  int load_eps;    // high corners read only if d > EPS   (deformable_conv.cu:254-261, 3-D :336-338)
  int atom_eps;    // high corners scattered only if d > EPS (mdeformable_conv.cu:285-293, 3-D)
  int range_gate;  // grad_offset only when -1 < p < size   (mdeformable_conv.cu:295)
  int with_bias;
  int modulated;
  // backward: 1 = add to the caller's gradient buffers (reference semantics, the default),
  // 0 = overwrite them (mdconv_set_accumulate); per-image gradients and weight gradients apart
  // because batch chunks after the first must add to grad_weight / grad_bias in either mode
  int acc_data, acc_w;
  // 1 = `input` is channels-last [B, spatial..., C] (mdconv_set_input_layout; 16-bit kernels only)
  int in_cl;
  // Group-padded channel layout of the native 16-bit path (hp_host.hip, hp_group_padded): deformable groups of 24 / 48 / ...
  // channels run as groups of cm_pad = 32 / 64 / ... channels -- C, Cg, Cdg above describe the PADDED problem the kernels see,
  // channel c of it is channel (c / cm_pad) * cm_real + c % cm_pad of the caller's C_caller-channel tensors when
  // c % cm_pad < cm_real, else padding (zero input, zero weights).  cm_pad == 0: channels are the caller's.
  int cm_pad, cm_real, C_caller;
};
// channel of the caller's input / weight / grad_input / grad_weight behind channel c of the kernels' rows; -1 = padding
__host__ __device__ __forceinline__ int caller_channel(const Geom &g, int c) {
  if (g.cm_pad == 0) return c < g.C ? c : -1;
  const int q = c / g.cm_pad, r = c - q * g.cm_pad;
  return (q < g.DG && r < g.cm_real) ? q * g.cm_real + r : -1;
}
__host__ __device__ __forceinline__ int caller_channels(const Geom &g) { return g.cm_pad ? g.C_caller : g.C; }

template <typename T> struct Acc { using type = float; };
template <> struct Acc<double> { using type = double; };

__device__ __forceinline__ float ld(const float *p) { return *p; }
__device__ __forceinline__ double ld(const double *p) { return *p; }
__device__ __forceinline__ float ld(const __half *p) { return __half2float(*p); }
__device__ __forceinline__ void st(float *p, float v) { *p = v; }
__device__ __forceinline__ void st(double *p, double v) { *p = v; }
__device__ __forceinline__ void st(__half *p, float v) { *p = __float2half(v); }
// bf16 storage type of the shape-generic kernels (the native 16-bit kernels use __bf16, hp_common.hpp)
struct bf16_t { unsigned short bits; };
__device__ __forceinline__ float bf16_bits_to_float(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ unsigned short float_to_bf16_bits(float v) {   // round to nearest even, NaN kept quiet
  const unsigned u = __float_as_uint(v);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float ld(const bf16_t *p) { return bf16_bits_to_float(p->bits); }
__device__ __forceinline__ void st(bf16_t *p, float v) { p->bits = float_to_bf16_bits(v); }

// Accumulating stores (the C ABI's backward entry points accumulate, include/mdconv.h).
__device__ __forceinline__ void atomic_add(float *p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add(double *p, double v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add(__half *p, float v) {
  // 16-bit atomics do not exist as scalar ops: CAS on the containing aligned dword.
  unsigned int *base = (unsigned int *)((uintptr_t)p & ~(uintptr_t)3);
  const bool hi = ((uintptr_t)p & 2) != 0;
  unsigned int old = *base, assumed;
  do {
    assumed = old;
    unsigned short h = hi ? (unsigned short)(assumed >> 16) : (unsigned short)(assumed & 0xffffu);
    const float f = __half2float(__ushort_as_half(h)) + v;
    const unsigned short nh = __half_as_ushort(__float2half(f));
    const unsigned int repl = hi ? ((assumed & 0x0000ffffu) | ((unsigned int)nh << 16))
                                 : ((assumed & 0xffff0000u) | nh);
    old = atomicCAS(base, assumed, repl);
  } while (old != assumed);
}

__device__ __forceinline__ void atomic_add(bf16_t *p, float v) {
  unsigned int *base = (unsigned int *)((uintptr_t)p & ~(uintptr_t)3);
  const bool hi = ((uintptr_t)p & 2) != 0;
  unsigned int old = *base, assumed;
  do {
    assumed = old;
    const unsigned short h = hi ? (unsigned short)(assumed >> 16) : (unsigned short)(assumed & 0xffffu);
    const unsigned short nh = float_to_bf16_bits(bf16_bits_to_float(h) + v);
    const unsigned int repl = hi ? ((assumed & 0x0000ffffu) | ((unsigned int)nh << 16))
                                 : ((assumed & 0xffff0000u) | nh);
    old = atomicCAS(base, assumed, repl);
  } while (old != assumed);
}

// Per-(tap, output pixel) sampling state, shared by every input channel of a deformable group.
template <int ND, typename A> struct TapCoef {
  A wl[ND];      // weight of the low side  (1-d), 0 if that side is outside the image
  A wh[ND];      // weight of the high side (d),   0 if outside (or, backward, d <= EPS & load_eps)
  A wha[ND];     // high-side weight used for the grad_input scatter (atom_eps flavour)
  A sl[ND];      // d(val)/d(p_a) factor of the low side:  -1 or 0
  A sh[ND];      // ... of the high side: +1 or 0
  int base;      // element index (inside one [S_i] plane) of the clamped low corner
  int delta[ND]; // index step low -> high per axis (0 when clamped)
  int last_lc;   // clamped low coordinate on the last (contiguous) axis
  bool inside;   // -1 < p_a < size_a on every axis
  bool vl[ND], vh[ND];   // the low / high side lies inside the image (the reference reads it)
  int low[ND];           // floor of the (clamped to [-2, size+1]) coordinate, NOT clamped to the image
};

// Decompose a flattened output pixel index into per-axis coordinates.
template <int ND> __device__ __forceinline__ void out_coords(const Geom &g, int pix, int *o) {
  if (ND == 2) {
    o[0] = pix / g.out_sz[1];
    o[1] = pix - o[0] * g.out_sz[1];
  } else {
    const int wl = g.out_sz[1] * g.out_sz[2];
    o[0] = pix / wl;
    const int r = pix - o[0] * wl;
    o[1] = r / g.out_sz[2];
    o[2] = r - o[1] * g.out_sz[2];
  }
}

template <int ND> __device__ __forceinline__ void tap_coords(const Geom &g, int tap, int *t) {
  if (ND == 2) {
    t[0] = tap / g.ksz[1];
    t[1] = tap - t[0] * g.ksz[1];
  } else {
    const int wl = g.ksz[1] * g.ksz[2];
    t[0] = tap / wl;
    const int r = tap - t[0] * wl;
    t[1] = r / g.ksz[2];
    t[2] = r - t[1] * g.ksz[2];
  }
}

// Build the sampling state from the ND offsets of one (tap, pixel).  `bwd` selects the
// backward gating flavours.
template <int ND, typename A>
__device__ __forceinline__ void make_tap(const Geom &g, const int *o, const int *t,
                                         const A *delta, bool bwd, TapCoef<ND, A> &tc) {
  int idx = 0;
  bool inside = true;
#pragma unroll
  for (int a = 0; a < ND; ++a) {
    const int size = g.in_sz[a];
    const A p = (A)(o[a] * g.stride[a] - g.pad[a] + t[a] * g.dil[a]) + delta[a];
    inside = inside && (p > (A)-1) && (p < (A)size);
    // clamp before the int conversion so wild offsets cannot overflow
    const A pc = p < (A)-2 ? (A)-2 : (p > (A)(size + 1) ? (A)(size + 1) : p);
    const A fl = floor(pc);
    const int low = (int)fl;
    const A d = pc - fl;
    const bool vl = (low >= 0) && (low <= size - 1);
    const bool vh = (low + 1 >= 0) && (low + 1 <= size - 1);
    const bool big = d > (A)MDCONV_EPS;
    const bool vh_load = vh && (!bwd || !g.load_eps || big);
    const bool vh_atom = vh && (!g.atom_eps || big);
    tc.vl[a] = vl;
    tc.vh[a] = vh_load;
    tc.low[a] = low;
    tc.wl[a] = vl ? (A)1 - d : (A)0;
    tc.wh[a] = vh_load ? d : (A)0;
    // grad_input scatter weight of the high side.  The modulated 2-D file writes it as `(p + 1 - high)` (mdeformable_conv.cu:288,
    // :292), which is `p - low` except that `p + 1` is rounded where it crosses a power of two -- visible on axes beyond
    // 2^15 pixels, where an fp32 coordinate has an ulp of 2^-8 pixels; restated literally so that grad_input agrees with the
    // reference there too (round 6; until then the kernels used `p - low` and differed by up to 2e-3 next to those columns)
    tc.wha[a] = vh_atom ? (g.range_gate ? (pc + (A)1) - (A)(low + 1) : d) : (A)0;
    tc.sl[a] = vl ? (A)-1 : (A)0;
    tc.sh[a] = vh_load ? (A)1 : (A)0;
    const int lc = low < 0 ? 0 : (low > size - 1 ? size - 1 : low);
    const int hc = low + 1 < 0 ? 0 : (low + 1 > size - 1 ? size - 1 : low + 1);
    int plane_stride = 1;
#pragma unroll
    for (int a2 = a + 1; a2 < ND; ++a2) plane_stride *= g.in_sz[a2];
    idx += lc * plane_stride;
    tc.delta[a] = (hc - lc) * plane_stride;
    if (a == ND - 1) tc.last_lc = lc;
  }
  tc.base = idx;
  tc.inside = inside;
}

// Corner ci: bit (ND-1-a) set <=> axis a takes the high side (the reference's v1..v4 / v1..v8
// order, deformable_conv3d.cu:21-44).
template <int ND, typename A>
__device__ __forceinline__ int corner_index(const TapCoef<ND, A> &tc, int ci) {
  int idx = tc.base;
#pragma unroll
  for (int a = 0; a < ND; ++a) idx += ((ci >> (ND - 1 - a)) & 1) ? tc.delta[a] : 0;
  return idx;
}
template <int ND, typename A>
__device__ __forceinline__ A corner_weight(const TapCoef<ND, A> &tc, int ci) {
  A w = (A)1;
#pragma unroll
  for (int a = 0; a < ND; ++a) w *= ((ci >> (ND - 1 - a)) & 1) ? tc.wh[a] : tc.wl[a];
  return w;
}
template <int ND, typename A>
__device__ __forceinline__ A corner_weight_atom(const TapCoef<ND, A> &tc, int ci) {
  A w = (A)1;
#pragma unroll
  for (int a = 0; a < ND; ++a) w *= ((ci >> (ND - 1 - a)) & 1) ? tc.wha[a] : tc.wl[a];
  return w;
}
// d(val)/d(p_axis) coefficient of corner ci.
template <int ND, typename A>
__device__ __forceinline__ A corner_dweight(const TapCoef<ND, A> &tc, int ci, int axis) {
  A w = (A)1;
#pragma unroll
  for (int a = 0; a < ND; ++a) {
    const bool hi = (ci >> (ND - 1 - a)) & 1;
    w *= (a == axis) ? (hi ? tc.sh[a] : tc.sl[a]) : (hi ? tc.wh[a] : tc.wl[a]);
  }
  return w;
}

// Paired corners: the two neighbours along the last (contiguous) axis are fetched with ONE 8-byte
// load from column cl = min(low, size-2), so a sample costs 2^(ND-1) gathers instead of 2^ND.
// Pair pi (bit (ND-2-a) set <=> axis a high, a < ND-1) gets element index idx[pi] (of its first
// element) and weights wx[pi] / wy[pi] for the two loaded values; clamped or out-of-image sides
// are handled purely through the weights.  Needs size of the last axis >= 2.
template <int ND, typename A>
__device__ __forceinline__ void make_pairs_f(const Geom &g, const TapCoef<ND, A> &tc, const A *fl,
                                             const A *fh, A scale, int (&idx)[1 << (ND - 1)],
                                             A (&wx)[1 << (ND - 1)], A (&wy)[1 << (ND - 1)]) {
  // fl[a] / fh[a]: factor of the low / high side on axis a (weights, or -1/+1 derivative factors)
  constexpr int L = ND - 1;
  const int lc = tc.last_lc, hc = tc.last_lc + tc.delta[L];
  const int cl = min(lc, g.in_sz[L] - 2);
  const A xw = (lc == cl ? fl[L] : (A)0) + (hc == cl ? fh[L] : (A)0);
  const A yw = (lc == cl + 1 ? fl[L] : (A)0) + (hc == cl + 1 ? fh[L] : (A)0);
#pragma unroll
  for (int pi = 0; pi < (1 << L); ++pi) {
    int id = tc.base - lc + cl;
    A w = scale;
#pragma unroll
    for (int a = 0; a < L; ++a) {
      const bool hi = (pi >> (L - 1 - a)) & 1;
      id += hi ? tc.delta[a] : 0;
      w *= hi ? fh[a] : fl[a];
    }
    idx[pi] = id;
    wx[pi] = w * xw;
    wy[pi] = w * yw;
  }
}
// A corner is READ by the reference iff it lies inside the image on every axis (mdeformable_conv.cu:9-34) and, in
// the backward of the files that gate their high loads, passes `d > EPS` (deformable_conv.cu:254-261): vl / vh.
template <int ND, typename A>
__device__ __forceinline__ bool corner_is_read(const TapCoef<ND, A> &tc, int ci) {
  bool ok = true;
#pragma unroll
  for (int a = 0; a < ND; ++a) ok = ok && (((ci >> (ND - 1 - a)) & 1) ? tc.vh[a] : tc.vl[a]);
  return ok;
}
// Which elements of the pairs the REFERENCE reads: an element stands for the low and / or the high side of the last
// axis (clamped sides collapse onto one column), and the reference loads a corner only if it lies inside the image on
// every axis (mdeformable_conv.cu:9-34, 256-267) and, in the files that gate their high loads, passes `d > EPS`
// (TapCoef::vl / vh).  Kernels that address corners one by one (channels-last gathers) park the others out of the
// buffer's range, so a non-finite value in a pixel the reference never touches cannot reach a result through 0 * Inf.
template <int ND, typename A>
__device__ __forceinline__ void make_pairs_read(const Geom &g, const TapCoef<ND, A> &tc, bool (&rx)[1 << (ND - 1)],
                                                bool (&ry)[1 << (ND - 1)]) {
  constexpr int L = ND - 1;
  const int lc = tc.last_lc, hc = tc.last_lc + tc.delta[L];
  const int cl = min(lc, g.in_sz[L] - 2);
  const bool xr = (lc == cl && tc.vl[L]) || (hc == cl && tc.vh[L]);
  const bool yr = (lc == cl + 1 && tc.vl[L]) || (hc == cl + 1 && tc.vh[L]);
#pragma unroll
  for (int pi = 0; pi < (1 << L); ++pi) {
    bool row = true;
#pragma unroll
    for (int a = 0; a < L; ++a) row = row && (((pi >> (L - 1 - a)) & 1) ? tc.vh[a] : tc.vl[a]);
    rx[pi] = row && xr;
    ry[pi] = row && yr;
  }
}
template <int ND, typename A>
__device__ __forceinline__ void make_pairs(const Geom &g, const TapCoef<ND, A> &tc, A scale,
                                           int (&idx)[1 << (ND - 1)], A (&wx)[1 << (ND - 1)],
                                           A (&wy)[1 << (ND - 1)]) {
  make_pairs_f<ND, A>(g, tc, tc.wl, tc.wh, scale, idx, wx, wy);
}
// same pairs, weights of d(val)/d(p_axis)
template <int ND, typename A>
__device__ __forceinline__ void make_pairs_d(const Geom &g, const TapCoef<ND, A> &tc, int axis,
                                             A (&wx)[1 << (ND - 1)], A (&wy)[1 << (ND - 1)]) {
  A fl[ND], fh[ND];
#pragma unroll
  for (int a = 0; a < ND; ++a) {
    fl[a] = a == axis ? tc.sl[a] : tc.wl[a];
    fh[a] = a == axis ? tc.sh[a] : tc.wh[a];
  }
  int idx[1 << (ND - 1)];
  make_pairs_f<ND, A>(g, tc, fl, fh, (A)1, idx, wx, wy);
}

// Scatter anchor of a SAMPLE for the inverted scatter map (hp_col2im.hip, mfma_csr3d.hip).  One list entry per
// sample instead of one per corner pair (2^(ND-1) times fewer entries and integer atomics): the
// entry is keyed by the sample's low corner in an EXTENDED index space -- low + 1 in [0, size] on
// every axis but the last, so that "target = low" and "target = low + 1" need no border cases --
// and by the pair column cl = min(low, size - 2) on the last axis (make_pairs_f); it carries the
// two column weights (mask folded in) and the low / high weights of the other axes.
template <int ND> struct SampleAnchor {
  int qa;                  // extended anchor index inside one (image, deformable group) segment
  float wx, wy;            // scatter weights (x mask) on columns cl and cl + 1
  float rl[ND - 1], rh[ND - 1];   // low / high scatter weights of the outer axes
  bool on;                 // the sample scatters anywhere at all (decided WITHOUT the mask)
};
__host__ __device__ inline int hp_anchor_space(const Geom &g) {
  int s = g.in_sz[g.nd - 1];
  for (int a = 0; a < g.nd - 1; ++a) s *= g.in_sz[a] + 1;
  return s;
}
template <int ND>
__device__ __forceinline__ void sample_anchor(const Geom &g, const TapCoef<ND, float> &tc, float m,
                                              SampleAnchor<ND> &sa) {
  constexpr int L = ND - 1;
  const int lc = tc.last_lc, hc = lc + tc.delta[L];   // the last axis has element stride 1
  const int cl = min(lc, g.in_sz[L] - 2);
  const float ux = (lc == cl ? tc.wl[L] : 0.f) + (hc == cl ? tc.wha[L] : 0.f);
  const float uy = (lc == cl + 1 ? tc.wl[L] : 0.f) + (hc == cl + 1 ? tc.wha[L] : 0.f);
  bool on = ux != 0.f || uy != 0.f;
  int q = 0;
#pragma unroll
  for (int a = 0; a < L; ++a) {
    const int e = tc.low[a] + 1;
    on = on && e >= 0 && e <= g.in_sz[a] && (tc.wl[a] != 0.f || tc.wha[a] != 0.f);
    q = q * (g.in_sz[a] + 1) + min(max(e, 0), g.in_sz[a]);
    sa.rl[a] = tc.wl[a];
    sa.rh[a] = tc.wha[a];
  }
  sa.qa = q * g.in_sz[L] + cl;
  sa.wx = ux * m;
  sa.wy = uy * m;
  sa.on = on;
}

// ---- exclusive scan of the scatter-list counters (shared by csr_scan_kernel and hp_csr_scan_kernel) ----
// cnt[seg][0..S) -> rowptr[seg][0..S]; grid (chunks of kScanChunk elements, segments), 256 threads.  A workgroup first sums
// everything before its chunk (the counters are L2-resident, at most S coalesced reads, eight loads in flight), then scans
// its chunk in ONE pass: thread t holds elements lo + 256 j + t (j = 0..7), the eight wave scans run as independent chains,
// one barrier publishes the 8 x 4 wave totals.  (Rounds 1-5 walked the chunk 256 elements at a time behind three barriers
// each: 45-70 us for the 8-64 workgroups of a cfg2 shard, on the critical path of the forked gather branch.)
constexpr int kScanChunk = 2048;
__device__ __forceinline__ void csr_scan_chunk(int S, const int *__restrict__ cnt, int *__restrict__ rowptr) {
  __shared__ int wtot[8][4];
  __shared__ int wpre[4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int seg = blockIdx.y;
  const int lo = blockIdx.x * kScanChunk, hi = min(lo + kScanChunk, S);
  const int *c = cnt + (int64_t)seg * S;
  int *rp = rowptr + (int64_t)seg * (S + 1);
  int v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int i = lo + j * 256 + tid;
    v[j] = i < hi ? c[i] : 0;
  }
  int pre = 0;
  for (int i0 = tid; i0 < lo; i0 += 8 * 256) {
    int a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = i0 + u * 256 < lo ? c[i0 + u * 256] : 0;
    pre += ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) pre += __shfl_xor(pre, d, 64);
  int x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = v[j];
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int y = __shfl_up(x[j], d, 64);
      if (lane >= d) x[j] += y;
    }
  }
  if (lane == 63) {
#pragma unroll
    for (int j = 0; j < 8; ++j) wtot[j][w] = x[j];
  }
  if (lane == 0) wpre[w] = pre;
  __syncthreads();
  int running = (wpre[0] + wpre[1]) + (wpre[2] + wpre[3]);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int t0 = wtot[j][0], t1 = wtot[j][1], t2 = wtot[j][2], t3 = wtot[j][3];
    const int woff = (w > 0 ? t0 : 0) + (w > 1 ? t1 : 0) + (w > 2 ? t2 : 0);
    const int i = lo + j * 256 + tid;
    if (i < hi) rp[i] = running + woff + x[j] - v[j];
    running += (t0 + t1) + (t2 + t3);
  }
  if (hi == S && tid == 0) rp[S] = running;
}

// ---- host-side helpers ------------------------------------------------------------------------
int fill_geom(const mdconv_desc *d, Geom *g);  // validates; returns MDCONV_* code
void set_error(const char *fmt, ...);
int check_launch(const char *what);

struct Tensors {
  const void *input, *weight, *bias, *offset, *mask, *grad_output;
  void *output, *grad_input, *grad_weight, *grad_bias, *grad_offset, *grad_mask;
};

// direct (VALU) path, any shape / dtype
int direct_forward(const Geom &g, int dtype, const Tensors &t, hipStream_t stream);
// parts: bit 0 = grad_input/grad_offset/grad_mask kernel, bit 1 = grad_weight/grad_bias kernel
int direct_backward(const Geom &g, int dtype, const Tensors &t, hipStream_t stream, int parts = 3);

}  // namespace mdconv
