// hp_common.hpp -- shared pieces of the native 16-bit (fp16 / bf16) kernels for gfx950.
//
// Half tensors never pass through fp32 copies (round 1 widened / narrowed every tensor): the
// kernels read __half / bf16 operands directly, keep coordinates, interpolation weights and every
// accumulator in fp32, and contract on v_mfma_f32_32x32x16_{f16,bf16} (16x the fp32 matrix
// rate), which turns the whole op from matrix-bound into texture-path (gather) bound.  Design:
//
//   * xt[b][q][Cp] -- channels-last copy of the input in its own 16-bit type (Cp = C_in rounded
//     up to 32, zero padded): one corner of 8 channels is ONE 16-byte load;
//   * a lane owns (pixel = lane & 31, channel octet = lane >> 5), which is exactly the B-operand
//     fragment of the 32x32x16 MFMA (N = pixel, K = 8 consecutive channels per half-wave): the
//     interpolated column values go from the gather registers straight into the matrix core, no
//     LDS round trip for the column operand at all;
//   * corners outside the image are never read: their buffer offset is parked out of range and the
//     hardware bounds check returns 0 (the reference's `if (h_low >= 0 ...)`,
//     mdeformable_conv.cu:9-34) -- also for non-finite border pixels.
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include "mdconv_common.hpp"
#include "mfma_tile.hpp"

namespace mdconv {

// developer aid (MDCONV_DEBUG_PLAN=1): grid size against the resident slots of a kernel instance, once per call site
template <typename K>
inline void hp_debug_plan(const char *name, K kernel, int threads, size_t lds, long blocks) {
  static const bool on = getenv("MDCONV_DEBUG_PLAN") != nullptr;
  if (!on) return;
  int n = 0, cus = 0, dev = 0;
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void *>(kernel), threads, lds);
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  fprintf(stderr, "[mdconv] %s: %ld workgroups of %d threads, %zu B LDS, %d resident per CU x %d CUs = %.2f rounds\n", name,
          blocks, threads, lds, n, cus, n > 0 && cus > 0 ? (double)blocks / ((double)n * cus) : 0.0);
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32;
struct U4 { u32 x, y, z, w; };   // 8 packed 16-bit elements

// buffer offset beyond every num_records (chunks stay below 0x7e000000 bytes) that stays positive
// when a lane adds its small in-row offset: loads return 0, stores are dropped
constexpr int kHpOob = 0x7f000000;

// 16-bit element types.  `Raw` = the in-memory type; everything is moved around as packed u32.
struct F16 {
  using Raw = _Float16;
  static __device__ __forceinline__ float lo(u32 p) { return (float)__builtin_bit_cast(f16x2, p)[0]; }
  static __device__ __forceinline__ float hi(u32 p) { return (float)__builtin_bit_cast(f16x2, p)[1]; }
  static __device__ __forceinline__ u32 pack(float a, float b) {
    const f16x2 v = {(_Float16)a, (_Float16)b};   // v_cvt_pk_f16_f32 (round to nearest even)
    return __builtin_bit_cast(u32, v);
  }
  static __device__ __forceinline__ float ldf(const Raw *p) { return (float)*p; }
  static __device__ __forceinline__ void stf(Raw *p, float v) { *p = (_Float16)v; }
  // acc + w * element(p, HI): one v_fma_mix_f32 (fp16 source, fp32 weight and accumulator)
  template <int HI> static __device__ __forceinline__ float mac(float acc, u32 p, float w) {
    float r;
    if (HI)
      asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(w), "v"(acc));
    else
      asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(w), "v"(acc));
    return r;
  }
  // acc + a.lo * b.lo + a.hi * b.hi in fp32 (v_dot2c_f32_f16)
  static __device__ __forceinline__ float dot2(float acc, u32 a, u32 b) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b), acc, false);
  }
  // acc (8 packed fp16) += w * x in PACKED fp16 arithmetic: one v_cvt_pk + four v_pk_fma_f16 per corner
  // against eight v_fma_mix_f32 (+ the final pack).  Used only for the column value that feeds GEMM-2
  // (col = sum of 2^ND products, rounded to fp16 for the matrix core anyway; grad_weight then averages
  // the extra rounding over every pixel of the batch) -- never for gradients written to the caller.
  static constexpr bool kPackedCol = true;
  static __device__ __forceinline__ void pk_mac8(U4 &acc, const U4 &x, float w) {
    const f16x2 w2 = {(_Float16)w, (_Float16)w};
    auto f = [&](u32 a, u32 v) {
      return __builtin_bit_cast(u32, __builtin_elementwise_fma(__builtin_bit_cast(f16x2, v), w2, __builtin_bit_cast(f16x2, a)));
    };
    acc.x = f(acc.x, x.x); acc.y = f(acc.y, x.y); acc.z = f(acc.z, x.z); acc.w = f(acc.w, x.w);
  }
  static __device__ __forceinline__ f32x16 mfma(const U4 &a, const U4 &b, const f32x16 &c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};
struct BF16 {
  using Raw = __bf16;
  static __device__ __forceinline__ float lo(u32 p) { return __builtin_bit_cast(float, p << 16); }
  static __device__ __forceinline__ float hi(u32 p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
  static __device__ __forceinline__ u32 pack(float a, float b) {
    const bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(u32, v);
  }
  static __device__ __forceinline__ float ldf(const Raw *p) { return (float)*p; }
  static __device__ __forceinline__ void stf(Raw *p, float v) { *p = (__bf16)v; }
  template <int HI> static __device__ __forceinline__ float mac(float acc, u32 p, float w) {
    return fmaf(w, HI ? hi(p) : lo(p), acc);
  }
  static __device__ __forceinline__ float dot2(float acc, u32 a, u32 b) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), acc, false);
  }
  static constexpr bool kPackedCol = false;
  static __device__ __forceinline__ void pk_mac8(U4 &, const U4 &, float) {}
  static __device__ __forceinline__ f32x16 mfma(const U4 &a, const U4 &b, const f32x16 &c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};

__device__ __forceinline__ U4 buf_load4u(rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(U4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store4u(rsrc_t r, int voff, int soff, const U4 &v) {
  typedef unsigned int u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned int))));
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
}

// streaming variants for rows that are written once / read once (aux bit 1 = nt): keeps them from evicting
// the gathered input rows from the L2.
__device__ __forceinline__ U4 buf_load4u_nt(rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(U4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 2));
}
__device__ __forceinline__ void buf_store4u_nt(rsrc_t r, int voff, int soff, const U4 &v) {
  typedef unsigned int u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned int))));
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 2);
}

// acc[j] += w * element j of the 8 packed values in v
template <typename T> __device__ __forceinline__ void mac8(float (&acc)[8], const U4 &v, float w) {
  acc[0] = T::template mac<0>(acc[0], v.x, w); acc[1] = T::template mac<1>(acc[1], v.x, w);
  acc[2] = T::template mac<0>(acc[2], v.y, w); acc[3] = T::template mac<1>(acc[3], v.y, w);
  acc[4] = T::template mac<0>(acc[4], v.z, w); acc[5] = T::template mac<1>(acc[5], v.z, w);
  acc[6] = T::template mac<0>(acc[6], v.w, w); acc[7] = T::template mac<1>(acc[7], v.w, w);
}
template <typename T> __device__ __forceinline__ U4 pack8(const float (&a)[8]) {
  U4 r;
  r.x = T::pack(a[0], a[1]); r.y = T::pack(a[2], a[3]); r.z = T::pack(a[4], a[5]); r.w = T::pack(a[6], a[7]);
  return r;
}
// s + sum over the 8 packed elements of a[j] * b[j] (fp32 accumulation).  NOTE: values that come
// straight out of an MFMA must reach inline asm (mac8) only through compiler-visible instructions
// such as these: hipcc does not pad the MFMA -> VALU read hazard for an asm statement.
template <typename T> __device__ __forceinline__ float dot8(float s, const U4 &a, const U4 &b) {
  s = T::dot2(s, a.x, b.x); s = T::dot2(s, a.y, b.y);
  s = T::dot2(s, a.z, b.z); s = T::dot2(s, a.w, b.w);
  return s;
}

// Per-(tap, pixel) sampling state for the channels-last gathers: element index of every corner
// inside one image of xt (or -1: the reference does not read that corner) and its interpolation
// weight (validity folded in; `bwd` selects the backward gating flavours of make_tap).
template <int ND> struct HpCorners {
  int idx[1 << ND];
  float w[1 << ND];
};
template <int ND>
__device__ __forceinline__ void hp_corners(const TapCoef<ND, float> &tc, HpCorners<ND> &hc) {
#pragma unroll
  for (int ci = 0; ci < (1 << ND); ++ci) {
    bool ok = true;
#pragma unroll
    for (int a = 0; a < ND; ++a) ok = ok && (((ci >> (ND - 1 - a)) & 1) ? tc.vh[a] : tc.vl[a]);
    hc.idx[ci] = ok ? corner_index<ND, float>(tc, ci) : -1;
    hc.w[ci] = corner_weight<ND, float>(tc, ci);
  }
}

constexpr int kPP = 40;   // LDS pitch (16-bit elements) of a 32-pixel grad_out row: 80 B, 16-byte aligned

// Sum of S[i] over the `sub` (a power of two, wave-uniform) lanes of a group, up to 16 lanes = one DPP
// row: quad_perm swaps (lane ^ 1, lane ^ 2), then row_half_mirror (lane i <-> 7 - i of its 8) and
// row_mirror (i <-> 15 - i), which pair lanes that already hold equal quad / half-row sums.
// ONE asm statement with scalar branches on `sub` inside: hipcc lowers the DPP builtin to mov +
// mov_dpp + add, and separate conditional statements cost a register copy per value and level;
// the s_nops cover the VALU-write -> DPP-read hazard (2 wait states), not padded for inline asm.
#define HP_DPP_LVL(MOD)                                                                           \
      "v_add_f32_dpp %0, %0, %0 " MOD " row_mask:0xf bank_mask:0xf\n\t"                           \
      "v_add_f32_dpp %1, %1, %1 " MOD " row_mask:0xf bank_mask:0xf\n\t"                           \
      "v_add_f32_dpp %2, %2, %2 " MOD " row_mask:0xf bank_mask:0xf\n\t"                           \
      "v_add_f32_dpp %3, %3, %3 " MOD " row_mask:0xf bank_mask:0xf\n\t"                           \
      "s_nop 1\n\t"
template <int NC> __device__ __forceinline__ void hp_dpp_sum(float (&S)[NC], int sub) {
#pragma unroll
  for (int o = 0; o < NC; o += 4)
    asm("s_nop 1\n\t"
        "s_cmp_lt_i32 %4, 2\n\t"
        "s_cbranch_scc1 Ldpp_end%=\n\t"
        HP_DPP_LVL("quad_perm:[1,0,3,2]")
        "s_cmp_lt_i32 %4, 4\n\t"
        "s_cbranch_scc1 Ldpp_end%=\n\t"
        HP_DPP_LVL("quad_perm:[2,3,0,1]")
        "s_cmp_lt_i32 %4, 8\n\t"
        "s_cbranch_scc1 Ldpp_end%=\n\t"
        HP_DPP_LVL("row_half_mirror")
        "s_cmp_lt_i32 %4, 16\n\t"
        "s_cbranch_scc1 Ldpp_end%=\n\t"
        HP_DPP_LVL("row_mirror")
        "Ldpp_end%=:"
        : "+v"(S[o]), "+v"(S[o + 1]), "+v"(S[o + 2]), "+v"(S[o + 3])
        : "s"(sub)
        : "scc");
}
#undef HP_DPP_LVL

typedef short s16x4 __attribute__((ext_vector_type(4)));
// 8 consecutive K values of this lane's matrix column from a row-major [K][N] LDS tile: two
// ds_read_b64_tr_b16 (4 rows each, `step` elements apart); `p` = this lane's piece of the block
template <typename Raw> __device__ __forceinline__ void lds_tr2(const Raw *p, int step, U4 &out) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p + step));
  struct P { s16x4 a, b; } pk = {a, b};
  out = __builtin_bit_cast(U4, pk);
}

// ---- tile order of the pixel-stationary kernels (hp_fwd2, hp_bwd3) ----
// A workgroup owns 128 output pixels = 4 wave segments of 32.  With linear order the 64 workgroups an
// XCD runs at a time (xcd_remap hands every XCD one contiguous run of tiles) cover two whole z-planes of
// a 3-D image; the input rows they gather for ONE tap then span 4-5 planes -- more than the XCD's 4 MB
// L2 at cfg5 (L2 hit 69 % forward, 65 % backward; a missing gather costs 4x the texture-path time of
// a hit, tools/ubench_gather16.hip).  Blocked order (3-D, row length 32 / 64 / 128, 8 | rows): tiles are
// numbered y-block of 8 rows slowest, then z, then the tiles of the 8 rows, so the same 64 workgroups
// cover 8 rows x all z -- a compact slab whose per-tap footprint fits the L2.
__host__ __device__ inline bool hp_blocked_ok(const Geom &g) {
  if (g.nd != 3) return false;
  const int W = g.out_sz[2];
  return (W == 32 || W == 64 || W == 128) && g.out_sz[1] % 8 == 0;
}
// first pixel (inside image b) of wave segment `wave` of tile sequence number `s`; b >= g.B: beyond the batch
__device__ __forceinline__ void hp_wave_segment(const Geom &g, int blocked, int s, int wave, int &b, int &pix0) {
  if (!blocked) {
    const int n0 = s * 128 + wave * 32;
    b = n0 / g.S_o;
    pix0 = n0 - b * g.S_o;
    return;
  }
  const int W = g.out_sz[2], H = g.out_sz[1], Z = g.out_sz[0];
  const int rt = 128 / W;               // rows per tile
  const int tpb = 8 / rt;               // tiles per (y-block, z)
  const int per_img = Z * (H / 8) * tpb;
  b = s / per_img;
  int r = s - b * per_img;
  const int yb = r / (Z * tpb);
  r -= yb * (Z * tpb);
  const int z = r / tpb, yt = r - z * tpb;
  const int seg = wave * 32;            // pixel offset inside the tile's rows
  const int y = yb * 8 + yt * rt + seg / W, x = seg % W;
  pix0 = (z * H + y) * W + x;
}

// ---- dimensions / workspace of the 16-bit path ----
struct HpDims {
  int Cp;           // C_in rounded up to 32: channel pitch of xt and of the grad_col rows
  int cblks;        // Cp / 32
  int Op;           // C_out rounded up to 32
  int oblks;        // Op / 32
  int OpL;          // rows of the backward kernel's grad_out tile in LDS (>= Op)
  // forward
  int MB;           // output-channel blocks (of 32) per workgroup: 1, 2, 4 or 8
  int oranges;      // workgroup rows along C_out = ceil(oblks / MB)
  int fwd_nmax;     // most output-channel blocks any 16-channel chunk can touch inside one row
  // backward
  int nks;          // GEMM-1 k-steps (16 output channels each) per 32-channel block
  int MB2;          // GEMM-2 output-channel blocks per 32-channel block
  int waves;        // waves per workgroup of the fused backward kernel = cblks (<= 8)
  int ranges;       // pixel ranges per tap of the fused backward kernel
  int tiles_per_range;
  int ranges_w;     // pixel ranges per tap of GEMM-2 (split-K of grad_weight, hp_gemm2.hip)
  int tiles_per_range_w;
  int max_ranges;   // upper bound of ranges / ranges_w for ANY batch size of this geometry (workspace sizing)
  int ntiles;       // 32-pixel tiles
  int blocked;      // pixel-stationary kernels: 1 = 128-pixel tiles in blocked order (hp_wave_segment), 0 = linear
};

}  // namespace mdconv
