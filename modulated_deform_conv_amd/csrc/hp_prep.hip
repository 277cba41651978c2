// hp_prep.hip -- layout passes of the native 16-bit path: channels-last input copy, MFMA-fragment
// weight packing (forward and backward), split-K reduction of grad_weight, grad_bias.
#include "hp_kernels.hpp"

namespace mdconv {

namespace {

int grid_for(int64_t total, int cap = 8192) {
  int64_t b = (total + 255) / 256;
  return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

// xt[b][q][Cp] = x[b][caller_channel(c)][q] (0 for padding channels): 64 x 64 tiles through LDS, 16-byte stores.
// Element type agnostic (moves 16-bit words).
__global__ __launch_bounds__(256) void hp_nchw_to_nhwc_kernel(Geom g, int Cp, int S,
                                                              const unsigned short *__restrict__ x,
                                                              unsigned short *__restrict__ xt) {
  __shared__ unsigned short t[64][66];
  const int b = blockIdx.z, c0 = blockIdx.y * 64, q0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
  for (int r = ty; r < 64; r += 4) {
    const int c = caller_channel(g, c0 + r), q = q0 + tx;
    t[r][tx] = (c >= 0 && q < S) ? x[((size_t)b * caller_channels(g) + c) * S + q] : (unsigned short)0;
  }
  __syncthreads();
  for (int item = threadIdx.x; item < 64 * 8; item += 256) {
    const int ql = item >> 3, oct = item & 7;
    const int q = q0 + ql, c = c0 + oct * 8;
    if (q < S && c < Cp) {
      U4 v;
      v.x = t[oct * 8 + 0][ql] | ((u32)t[oct * 8 + 1][ql] << 16);
      v.y = t[oct * 8 + 2][ql] | ((u32)t[oct * 8 + 3][ql] << 16);
      v.z = t[oct * 8 + 4][ql] | ((u32)t[oct * 8 + 5][ql] << 16);
      v.w = t[oct * 8 + 6][ql] | ((u32)t[oct * 8 + 7][ql] << 16);
      *reinterpret_cast<U4 *>(xt + ((size_t)b * S + q) * Cp + c) = v;
    }
  }
}

// The same copy with 8-byte loads along q (4 pixels of one channel per lane: 4 load instructions per thread instead
// of 16 two-byte ones) for S a multiple of 4 and an 8-byte aligned source; LDS tile and the 16-byte stores as above.
// Picked for rows of 512 bytes (cfg3, C = 256), where the transposing-read variant below loses.
__global__ __launch_bounds__(256) void hp_nchw_to_nhwc_q4_kernel(Geom g, int Cp, int S,
                                                                 const unsigned short *__restrict__ x,
                                                                 unsigned short *__restrict__ xt) {
  __shared__ __attribute__((aligned(8))) unsigned short t[64][68];   // pitch 136 B: 8-byte aligned rows
  const int b = blockIdx.z, c0 = blockIdx.y * 64, q0 = blockIdx.x * 64;
  const int u = threadIdx.x & 15, r0 = threadIdx.x >> 4;   // pixel quad, channel row (16 rows per pass)
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int r = r0 + 16 * pass, c = caller_channel(g, c0 + r), q = q0 + u * 4;
    uint2 v = make_uint2(0u, 0u);
    if (c >= 0 && q < S) v = *reinterpret_cast<const uint2 *>(x + ((size_t)b * caller_channels(g) + c) * S + q);   // S % 4 == 0: whole quad inside
    *reinterpret_cast<uint2 *>(&t[r][u * 4]) = v;
  }
  __syncthreads();
  for (int item = threadIdx.x; item < 64 * 8; item += 256) {
    const int ql = item >> 3, oct = item & 7;
    const int q = q0 + ql, c = c0 + oct * 8;
    if (q < S && c < Cp) {
      U4 v;
      v.x = t[oct * 8 + 0][ql] | ((u32)t[oct * 8 + 1][ql] << 16);
      v.y = t[oct * 8 + 2][ql] | ((u32)t[oct * 8 + 3][ql] << 16);
      v.z = t[oct * 8 + 4][ql] | ((u32)t[oct * 8 + 5][ql] << 16);
      v.w = t[oct * 8 + 6][ql] | ((u32)t[oct * 8 + 7][ql] << 16);
      *reinterpret_cast<U4 *>(xt + ((size_t)b * S + q) * Cp + c) = v;
    }
  }
}

// The same copy for S a multiple of 8 (and a 16-byte aligned source): 16-byte loads along q (2 per thread instead of
// 16 two-byte ones), rows written to LDS as loaded, and the transpose done by the LDS itself -- ds_read_b64_tr_b16
// hands lane i of a 16-lane group column i of a 4 x 16 block, two of them are the 8 channels of one output pixel.
// Row pitch 80 elements: the four rows of a block start 8 banks apart.  The 16-byte stores of neighbouring lanes
// land one pixel row apart, which pays for rows of 256 bytes (cfg5, C = 128: 95 -> 61 us) and not for rows of 512
// (cfg3, C = 256: 36 -> 41 us) -- the launcher picks by row length.  (Vector loads with the two-byte LDS reads of
// the kernel above: 16-byte aligned rows put a pixel's 8 channel octets in one bank, 49 / 82 us.)
__global__ __launch_bounds__(256) void hp_nchw_to_nhwc_vec_kernel(Geom g, int Cp, int S,
                                                                  const unsigned short *__restrict__ x,
                                                                  unsigned short *__restrict__ xt) {
  constexpr int P = 80;
  __shared__ __attribute__((aligned(16))) unsigned short t[64 * P];
  const int b = blockIdx.z, c0 = blockIdx.y * 64, q0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (tid >> 3) + 32 * i, u = tid & 7;
    const int c = caller_channel(g, c0 + row), q = q0 + u * 8;
    U4 v = {0, 0, 0, 0};
    if (c >= 0 && q < S) v = *reinterpret_cast<const U4 *>(x + ((size_t)b * caller_channels(g) + c) * S + q);
    *reinterpret_cast<U4 *>(t + row * P + u * 8) = v;
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6, i16 = lane & 15;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int combo = pass * 16 + wave * 4 + (lane >> 4);   // (pixel block of 16, channel octet)
    const int oct = combo & 7, qb = combo >> 3;
    U4 v;
    lds_tr2(t + (oct * 8 + (i16 >> 2)) * P + qb * 16 + (i16 & 3) * 4, 4 * P, v);
    const int q = q0 + qb * 16 + i16, c = c0 + oct * 8;
    if (q < S && c < Cp) *reinterpret_cast<U4 *>(xt + ((size_t)b * S + q) * Cp + c) = v;
  }
}

// forward A operand: wpf[tap][chunk][oblk][lane][8] = W[o = oblk*32 + (lane&31)]
//                                                      [c = chunk*16 + 8*(lane>>5) + j][tap]
// dense block-diagonal over conv groups (0 where o and c belong to different groups, or padding).
// element index of weight[o][c][tap] in the caller's [O][C_in / groups][K] tensor for channel c of the kernels' rows; -1 where
// o and c belong to different conv groups or either is padding (the group-padded layout has one conv group)
__device__ __forceinline__ int64_t hp_weight_index(const Geom &g, int o, int c, int tap) {
  if (o >= g.O) return -1;
  if (g.cm_pad) {
    const int cc = caller_channel(g, c);
    return cc < 0 ? -1 : ((int64_t)o * g.C_caller + cc) * g.K + tap;
  }
  return (c < g.C && o / g.Og == c / g.Cg) ? ((int64_t)o * g.Cg + (c % g.Cg)) * g.K + tap : -1;
}
__device__ __forceinline__ unsigned short hp_weight_or_zero(const Geom &g, const unsigned short *__restrict__ w, int o, int c,
                                                           int tap) {
  const int64_t e = hp_weight_index(g, o, c, tap);
  return e < 0 ? (unsigned short)0 : w[e];
}
__device__ void hp_ctab_fill(const Geom &g, const HpDims &hd, int2 *__restrict__ ctab);
__device__ __forceinline__ int4 hp_btab_entry(const Geom &g, int cblk);
// (block 0 also writes the chunk table the forward kernel reads: one launch less than a table kernel of its own)
__global__ __launch_bounds__(256) void hp_pack_fwd_kernel(Geom g, HpDims hd,
                                                          const unsigned short *__restrict__ w,
                                                          U4 *__restrict__ wpf, int2 *__restrict__ ctab) {
  if (blockIdx.x == 0) hp_ctab_fill(g, hd, ctab);
  const int nchunks = hd.Cp / 16;
  const int64_t total = (int64_t)g.K * nchunks * hd.oblks * 64;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t r = i;
    const int lane = (int)(r & 63); r >>= 6;
    const int oblk = (int)(r % hd.oblks); r /= hd.oblks;
    const int chunk = (int)(r % nchunks);
    const int tap = (int)(r / nchunks);
    const int o = oblk * 32 + (lane & 31);
    const int cb = chunk * 16 + 8 * (lane >> 5);
    unsigned short e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cb + j;
      e[j] = hp_weight_or_zero(g, w, o, c, tap);
    }
    U4 v;
    v.x = e[0] | ((u32)e[1] << 16); v.y = e[2] | ((u32)e[3] << 16);
    v.z = e[4] | ((u32)e[5] << 16); v.w = e[6] | ((u32)e[7] << 16);
    wpf[i] = v;
  }
}

// per workgroup row (`orange`) and 16-channel chunk: the range of the row's output-channel blocks
// that can be non-zero (relative to the row's first block); entry [nchunks] = the row's chunk range
__device__ void hp_ctab_fill(const Geom &g, const HpDims &hd, int2 *__restrict__ ctab) {
  const int nchunks = hd.Cp / 16;
  for (int orange = threadIdx.x; orange < hd.oranges; orange += blockDim.x) {
    int2 *ct = ctab + orange * (nchunks + 1);
    const int b_lo = orange * hd.MB, b_hi = min(b_lo + hd.MB, hd.oblks) - 1;
    int ch_lo = nchunks, ch_hi = 0;
    for (int ch = 0; ch < nchunks; ++ch) {
      int lo = 0, n = 0;
      if (ch * 16 < g.C) {
        const int g_lo = (ch * 16) / g.Cg, g_hi = min(ch * 16 + 15, g.C - 1) / g.Cg;
        const int ob_lo = max((g_lo * g.Og) / 32, b_lo);
        const int ob_hi = min(((g_hi + 1) * g.Og - 1) / 32, b_hi);
        if (ob_hi >= ob_lo) {
          lo = ob_lo - b_lo;
          n = ob_hi - ob_lo + 1;
          ch_lo = min(ch_lo, ch);
          ch_hi = max(ch_hi, ch + 1);
        }
      }
      ct[ch] = make_int2(lo, n);
    }
    ct[nchunks] = make_int2(ch_lo, ch_hi);
  }
}

// channel of MFMA row i of the backward A operand: a lane of the 32x32 accumulator then owns 16
// CONSECUTIVE channels (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)  ->  channel 16*(lane>>5) + reg)
__host__ __device__ inline int hp_sigma(int i) { return 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3); }

// backward A operand (W^T): wpb[tap][cblk][ks][lane][8] = W[o = o_base(cblk) + ks*16 + 8*(lane>>5) + j]
//                                                           [c = cblk*32 + sigma(lane&31)][tap]
// (the per-block output base is computed in place; block 0 writes the table the later kernels read)
__global__ __launch_bounds__(256) void hp_pack_bwd_kernel(Geom g, HpDims hd, int4 *__restrict__ btab,
                                                          const unsigned short *__restrict__ w,
                                                          U4 *__restrict__ wpb) {
  if (blockIdx.x == 0)
    for (int cb = threadIdx.x; cb < hd.cblks; cb += blockDim.x) btab[cb] = hp_btab_entry(g, cb);
  const int64_t total = (int64_t)g.K * hd.cblks * hd.nks * 64;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t r = i;
    const int lane = (int)(r & 63); r >>= 6;
    const int ks = (int)(r % hd.nks); r /= hd.nks;
    const int cblk = (int)(r % hd.cblks);
    const int tap = (int)(r / hd.cblks);
    const int c = cblk * 32 + hp_sigma(lane & 31);
    const int ob = hp_btab_entry(g, cblk).x + ks * 16 + 8 * (lane >> 5);
    unsigned short e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int o = ob + j;
      e[j] = hp_weight_or_zero(g, w, o, c, tap);
    }
    U4 v;
    v.x = e[0] | ((u32)e[1] << 16); v.y = e[2] | ((u32)e[3] << 16);
    v.z = e[4] | ((u32)e[5] << 16); v.w = e[6] | ((u32)e[7] << 16);
    wpb[i] = v;
  }
}

// per 32-channel block: first output channel (32-aligned) of the groups its channels belong to
__device__ __forceinline__ int4 hp_btab_entry(const Geom &g, int cblk) {
  const int c_lo = min(cblk * 32, g.C - 1), c_hi = min(cblk * 32 + 31, g.C - 1);
  const int o_lo = (c_lo / g.Cg) * g.Og, o_hi = (c_hi / g.Cg + 1) * g.Og;
  const int base = o_lo / 32 * 32;
  return make_int4(base, (o_hi - base + 31) / 32, 0, 0);
}

// grad_weight[o][c][tap] (+)= sum over the pixel ranges of part[tap][range][cblk][ob][lane][16]
// (the 32x32 fp32 accumulator blocks of the fused backward kernel, D[i = o][j = c])
template <typename T>
__global__ __launch_bounds__(256) void hp_reduce_gw_kernel(Geom g, HpDims hd, int ranges, const int4 *__restrict__ btab,
                                                           const float *__restrict__ part,
                                                           typename T::Raw *__restrict__ gw,
                                                           float *__restrict__ gw32, int first, int last) {
  const int64_t total = (int64_t)g.K * hd.cblks * hd.MB2 * 1024;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t r = i;
    const int reg = (int)(r & 15); r >>= 4;
    const int lane = (int)(r & 63); r >>= 6;
    const int ob = (int)(r % hd.MB2); r /= hd.MB2;
    const int cblk = (int)(r % hd.cblks);
    const int tap = (int)(r / hd.cblks);
    const int o = btab[cblk].x + ob * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    const int c = cblk * 32 + (lane & 31);
    const int64_t e = hp_weight_index(g, o, c, tap);
    if (e >= 0) {
      const int64_t per_range = (int64_t)hd.cblks * hd.MB2 * 1024;
      const float *p = part + ((int64_t)tap * ranges) * per_range + ((int64_t)cblk * hd.MB2 + ob) * 1024 +
                       lane * 16 + reg;
      float s = 0.f;   // four partials in flight, fixed order (a plain loop left one dependent load per range)
      int k = 0;
      for (; k + 4 <= ranges; k += 4) {
        const float a0 = p[(int64_t)k * per_range], a1 = p[(int64_t)(k + 1) * per_range];
        const float a2 = p[(int64_t)(k + 2) * per_range], a3 = p[(int64_t)(k + 3) * per_range];
        s += (a0 + a1) + (a2 + a3);
      }
      for (; k < ranges; ++k) s += p[(int64_t)k * per_range];
      // calls cut into batch chunks keep the running sum in fp32 (gw32) and round ONCE, after the
      // last chunk, like the single-chunk path
      if (gw32) {
        if (!first) s += gw32[e];
        if (!last) { gw32[e] = s; continue; }
      }
      typename T::Raw *dst = gw + e;
      T::stf(dst, g.acc_w ? T::ldf(dst) + s : s);
    }
  }
}

// grad_bias[o] (+)= sum over (b, pix) of grad_out[b][o][pix].  One workgroup of 1024 threads per output channel; a row
// [S_o] of (image, channel) is read with 16-byte loads (eight elements), four in flight per thread, the few elements before
// the row's first / after its last 16-byte boundary one by one.  Per-thread fp32 partials and a fixed tree: bit-reproducible
// from run to run.  (Until round 5: 256 threads reading one element each per step -- 2 bytes per lane and one load in flight:
// +0.14 ms on the cfg3 backward, +0.88 ms on cfg5 with bias; profiles/r05_experiments.md 28.)
template <typename T>
__device__ __forceinline__ float sum8(const U4 &v) {
  return ((T::lo(v.x) + T::hi(v.x)) + (T::lo(v.y) + T::hi(v.y))) + ((T::lo(v.z) + T::hi(v.z)) + (T::lo(v.w) + T::hi(v.w)));
}
constexpr int kBiasThreads = 1024;
template <typename T>
__global__ __launch_bounds__(kBiasThreads) void hp_grad_bias_kernel(Geom g, const typename T::Raw *__restrict__ go,
                                                                    typename T::Raw *__restrict__ gb) {
  __shared__ float red[kBiasThreads];
  const int o = blockIdx.x, tid = threadIdx.x;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (int b = 0; b < g.B; ++b) {
    const typename T::Raw *src = go + ((int64_t)b * g.O + o) * g.S_o;
    int head = (int)((8 - (((uintptr_t)src >> 1) & 7)) & 7);   // elements before the first 16-byte boundary
    if (head > g.S_o) head = g.S_o;
    const int nvec = (g.S_o - head) >> 3;
    const int tail0 = head + nvec * 8;
    if (tid < head) s2 += T::ldf(src + tid);
    if (tid < g.S_o - tail0) s3 += T::ldf(src + tail0 + tid);
    const U4 *v = reinterpret_cast<const U4 *>(src + head);
    int i = tid;
    for (; i + 3 * kBiasThreads < nvec; i += 4 * kBiasThreads) {
      const U4 a0 = v[i], a1 = v[i + kBiasThreads], a2 = v[i + 2 * kBiasThreads], a3 = v[i + 3 * kBiasThreads];
      s0 += sum8<T>(a0);
      s1 += sum8<T>(a1);
      s2 += sum8<T>(a2);
      s3 += sum8<T>(a3);
    }
    for (; i < nvec; i += kBiasThreads) s0 += sum8<T>(v[i]);
  }
  red[tid] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  for (int d = kBiasThreads / 2; d > 0; d >>= 1) {
    if (tid < d) red[tid] += red[tid + d];
    __syncthreads();
  }
  if (tid == 0) T::stf(gb + o, g.acc_w ? T::ldf(gb + o) + red[0] : red[0]);
}

}  // namespace

int hp_nchw_to_nhwc(const Geom &g, const HpDims &hd, const void *x, void *xt, hipStream_t stream) {
  const dim3 grid((g.S_i + 63) / 64, (hd.Cp + 63) / 64, g.B);
  if (hd.Cp <= 128 && g.S_i % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
    hipLaunchKernelGGL(hp_nchw_to_nhwc_vec_kernel, grid, dim3(256), 0, stream, g, hd.Cp, g.S_i,
                       (const unsigned short *)x, (unsigned short *)xt);
  else if (g.S_i % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 7) == 0)
    hipLaunchKernelGGL(hp_nchw_to_nhwc_q4_kernel, grid, dim3(256), 0, stream, g, hd.Cp, g.S_i,
                       (const unsigned short *)x, (unsigned short *)xt);
  else
    hipLaunchKernelGGL(hp_nchw_to_nhwc_kernel, grid, dim3(256), 0, stream, g, hd.Cp, g.S_i,
                       (const unsigned short *)x, (unsigned short *)xt);
  return check_launch("hp_nchw_to_nhwc");
}

int hp_pack_fwd_weights(const Geom &g, const HpDims &hd, int dtype, const void *w, void *wpf,
                        int2 *ctab, hipStream_t stream) {
  const int64_t total = (int64_t)g.K * (hd.Cp / 16) * hd.oblks * 64;
  hipLaunchKernelGGL(hp_pack_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, g, hd,
                     (const unsigned short *)w, (U4 *)wpf, ctab);
  return check_launch("hp_pack_fwd");
}

int hp_pack_bwd_weights(const Geom &g, const HpDims &hd, int dtype, const void *w, void *wpb,
                        int4 *btab, hipStream_t stream) {
  const int64_t total = (int64_t)g.K * hd.cblks * hd.nks * 64;
  hipLaunchKernelGGL(hp_pack_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, g, hd, btab,
                     (const unsigned short *)w, (U4 *)wpb);
  return check_launch("hp_pack_bwd");
}

int hp_reduce_grad_weight(const Geom &g, const HpDims &hd, int ranges, int dtype, const float *part,
                          const int4 *btab, void *grad_weight, float *gw32, bool first, bool last,
                          hipStream_t stream) {
  const int64_t total = (int64_t)g.K * hd.cblks * hd.MB2 * 1024;
  if (dtype == MDCONV_F16)
    hipLaunchKernelGGL((hp_reduce_gw_kernel<F16>), dim3(grid_for(total)), dim3(256), 0, stream, g, hd,
                       ranges, btab, part, (_Float16 *)grad_weight, gw32, first ? 1 : 0, last ? 1 : 0);
  else
    hipLaunchKernelGGL((hp_reduce_gw_kernel<BF16>), dim3(grid_for(total)), dim3(256), 0, stream, g, hd,
                       ranges, btab, part, (__bf16 *)grad_weight, gw32, first ? 1 : 0, last ? 1 : 0);
  return check_launch("hp_reduce_gw");
}

int hp_grad_bias(const Geom &g, int dtype, const void *grad_output, void *grad_bias,
                 hipStream_t stream) {
  if (dtype == MDCONV_F16)
    hipLaunchKernelGGL((hp_grad_bias_kernel<F16>), dim3(g.O), dim3(kBiasThreads), 0, stream, g,
                       (const _Float16 *)grad_output, (_Float16 *)grad_bias);
  else
    hipLaunchKernelGGL((hp_grad_bias_kernel<BF16>), dim3(g.O), dim3(kBiasThreads), 0, stream, g,
                       (const __bf16 *)grad_output, (__bf16 *)grad_bias);
  return check_launch("hp_grad_bias");
}

}  // namespace mdconv
