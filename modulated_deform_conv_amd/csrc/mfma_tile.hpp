// mfma_tile.hpp -- shared pieces of the MFMA implicit-GEMM kernels (fp32, gfx950).
#pragma once
#include "mdconv_common.hpp"

namespace mdconv {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBK = 16;  // K-chunk (channels of one tap) per LDS stage

// Padded dimensions of the packed weight copies kept in the workspace.
struct PackDims {
  int Cgp;  // C_in/groups rounded up to 2*kBK
  int Ogp;  // C_out/groups rounded up to BM
  int BM;   // output-channel tile of the forward kernel (256 / 128 / 64)
};

inline PackDims pack_dims(const Geom &g) {
  PackDims pd;
  pd.BM = g.Og > 128 ? 256 : (g.Og > 64 ? 128 : 64);
  pd.Cgp = (g.Cg + 2 * kBK - 1) / (2 * kBK) * (2 * kBK);  // even chunk count (2x unrolled K loops)
  pd.Ogp = (g.Og + pd.BM - 1) / pd.BM * pd.BM;
  return pd;
}

// Map the hardware's round-robin block->XCD placement (block b runs on XCD b % 8,
// MI355X_MICROARCH.md) to contiguous tile ranges per XCD so neighbouring tiles share an L2.
// Bijective for any n (cdna_hip_programming.md, "XCD swizzle must be bijective").  Speed only.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
  const int q = n >> 3, r = n & 7;
  const int xcd = bid & 7, slot = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// ---- raw buffer loads: base in an SGPR resource, per-lane byte offset in `voff`, wave-uniform
// byte offset in `soff` (lives in an SGPR), so the address costs no VALU instruction.  Out of
// range offsets return 0 instead of faulting (num_records bound). ----
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void *p, size_t bytes) {
  const unsigned n = bytes > 0xfffffff0ull ? 0xfffffff0u : (unsigned)bytes;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)n, 0x00020000);
}
__device__ __forceinline__ float buf_load(rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
struct F2bits { float x, y; };
__device__ __forceinline__ float2 buf_load2(rsrc_t r, int voff, int soff) {
  const F2bits f = __builtin_bit_cast(F2bits, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
  return make_float2(f.x, f.y);
}
struct F4bits { float x, y, z, w; };
__device__ __forceinline__ float4 buf_load4(rsrc_t r, int voff, int soff) {
  // The builtin returns an opaque 128-bit value: assigning it to an int4 vector SPLATS it
  // (every component = the first dword, and hipcc then narrows the load to one dword), so
  // reinterpret the bits through a struct instead.
  // (its type is a GCC-style vector of 4 unsigned; bit_cast is the safe way out)
  const F4bits f = __builtin_bit_cast(F4bits, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
  return make_float4(f.x, f.y, f.z, f.w);
}

template <int AUX = 0>
__device__ __forceinline__ void buf_store4(rsrc_t r, int voff, int soff, float a, float b, float c,
                                           float d) {
  typedef unsigned int u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned int))));
  const u32x4 v = {__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b),
                   __builtin_bit_cast(unsigned, c), __builtin_bit_cast(unsigned, d)};
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, AUX);   // AUX 2 = nt (streaming)
}

// Dimensions / workspace layout of the MFMA backward (fp32, groups == 1).
struct BwdDims {
  int Np;               // B*S_o rounded up to 32 (even number of 16-pixel chunks)
  int cl;               // GEMM-2 gathers from the channels-last input copy (mfma_bwd_weight_cl.hip)
  int wtile;            // GEMM-2 workgroup tile: 0 = 256(o) x 32(c), 1 = 64 x 64 (C_out <= 64);
                        // channels-last: 1 = 64 x 64, 2 = 128 x 64, 3 = 256 x 64
  int OgpB, mblks, mtiles;   // C_out rounded up to the tile rows; /32; / tile rows
  int Cp, cblks;        // C_in rounded up to the tile channels; / tile channels
  int splits, pairs_per_split;   // split-K of the grad_weight GEMM over pixel-chunk pairs
  int ochunks;          // C_out rounded up to 32, /16 (even): K chunks of GEMM-1
  int waves_c, cblks_q; // GEMM-1: waves along channels (4/2/1), 32-channel blocks of wq
  // workspace byte offsets
  int bias_tiles;       // pixel tiles of GEMM-1 = rows of the grad_bias partial sums
  int red_floats;       // GEMM-1: floats of the grad_offset / grad_mask reduction buffer in LDS
  int tap_group;        // GEMM-1: taps per flush of that buffer (9; 3 where LDS is short)
  int cl_drain;         // GEMM-1 drains through the channels-last copy (line-wide gathers)
  int sample_keyed;     // scatter lists: 1 = one entry per SAMPLE (3-D, mfma_csr3d.hip), 0 = per corner pair
  int S_e;              // list heads per (image, deformable group): anchor space (3-D) or S_i
  size_t off_wq, off_ga, off_table, off_part, off_gcol, off_cnt, off_rowptr, off_entries, off_bias,
      off_xt, off_sums, off_bstage, off_end;   // off_sums: per-anchor partial sums of the two-pass 3-D gather (0 bytes otherwise)
  int two_pass;         // 3-D grad_input gather: 1 = per-anchor partial sums + stencil (mfma_csr3d.hip), 0 = block walk
};
BwdDims bwd_dims(const Geom &g);

#ifdef __HIPCC__
// xt[b][q0 + r][c0 + tx] = x[b][c0 + r][q0 + tx] for one 32 x 32 tile through LDS (256 threads; C is a multiple of 32)
__device__ __forceinline__ void nchw_to_nhwc_tile(float (*t)[33], int C, int S, const float *__restrict__ x,
                                                  float *__restrict__ xt, int b, int c0, int q0) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 8 rows per pass
  const float *src = x + ((size_t)b * C + c0) * S;
  float *dst = xt + ((size_t)b * S + q0) * C + c0;
#pragma unroll
  for (int r = ty; r < 32; r += 8) t[r][tx] = (q0 + tx < S) ? src[(size_t)r * S + q0 + tx] : 0.f;
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8)
    if (q0 + r < S) dst[(size_t)r * C + tx] = t[tx][r];
}
#endif

// ---- internal entry points (fp32) ----
// wp : forward A operand in MFMA-fragment order, so a wave reads its 32x8 fragment with ONE fully
//      coalesced 16-byte-per-lane load:
//        wp[g][tap][cchunk][mblk][q][lane][s] = W[g*Og + mblk*32 + (lane&31)]
//                                                [cchunk*16 + 8*q + 4*(lane>>5) + s][tap]
//      (cchunk < Cgp/16, mblk < Ogp/32, q < 2, s < 4; zero padded).  MFMA step (q, s) multiplies
//      k = cchunk*16 + 8q + s (lanes 0-31) and k + 4 (lanes 32-63).
// wq : [G][K][Ogp][Cgp]  (tap-major, input channel contiguous)   -- backward GEMM-1 operand
int pack_weights_f32(const Geom &g, const PackDims &pd, const float *weight, float *wp, float *wq,
                     hipStream_t stream);
// part: scratch of fwd_tail_bytes(g) for the tap-range partials of the last dispatch round (nullptr = no tail split)
size_t fwd_tail_bytes(const Geom &g);
// tail plan of a forward tile grid and the reduction of its tap-range partials (mfma_fwd.hip; shared with the channels-last
// forward of mfma_fwd_cl.hip): see fwd_tail_plan there
constexpr int kTailMaxPerCu = 5;   // resident workgroups per CU the tail plan and its scratch are sized for
void fwd_tail_plan(const Geom &g, int tiles, int slots, int *full_tiles, int *ways, int *n_hi);
int fwd_tail_reduce_launch(int BM, int BN, const Geom &g, const float *part, const float *bias, float *output, int ntm,
                           int tail_tiles, int full_tiles, int ways, int n_hi, hipStream_t stream);
int mfma_forward_f32(const Geom &g, const PackDims &pd, const Tensors &t, const float *wp, float *part,
                     hipStream_t stream);
// channels-last gathers (mfma_fwd_cl.hip): xt = scratch for the NHWC copy of the input
bool fwd_channels_last(const Geom &g);
size_t fwd_cl_bytes(const Geom &g);
int mfma_forward_cl_f32(const Geom &g, const PackDims &pd, const Tensors &t, const float *wp, float *part,
                        float *xt, hipStream_t stream);
int mfma_bwd_weight_f32(const Geom &g, const BwdDims &bd, const Tensors &t, const float *ga,
                        const int *table, float *part, const float *bias_part, const float *xt,
                        hipStream_t stream);
int mfma_bwd_weight_cl_launch(const Geom &g, const BwdDims &bd, const float *xt, const float *ga,
                              const int *table, float *part, hipStream_t stream);
// resident workgroups per CU of the GEMM-2 instance a shape selects (hipOccupancy, cached); device_cus() = CUs of
// the current device (256 on MI355X; the same figure without a device, for host-only callers)
int mfma_bwd_weight_cl_occupancy(int nd, bool padn, int wtile);
int mfma_bwd_weight_occupancy(int nd, bool padn, int wtile);
int device_cus();
bool bwd_channels_last(const Geom &g);
int nchw_to_nhwc_f32(const Geom &g, const float *x, float *xt, hipStream_t stream);
int pack_wq_f32(const Geom &g, const BwdDims &bd, const float *weight, float *wq, hipStream_t stream);
// pack_wq + counter clearing + (xt != nullptr) the channels-last input copy as ONE launch: three dependent 5-40 us
// kernels in front of GEMM-1 cost two launch gaps of 6-22 us on top of their own time (kernel trace, round 4)
int bwd_prep_f32(const Geom &g, const BwdDims &bd, const float *weight, float *wq, int *cnt, const float *x,
                 float *xt, hipStream_t stream);
// grad_bias from the per-tile partial sums GEMM-1 left (two ordered stages; `stage` = grad_bias_stage_bytes scratch)
size_t grad_bias_stage_bytes(const Geom &g);
int grad_bias_f32(const Geom &g, const BwdDims &bd, const float *bias_part, float *stage, float *grad_bias,
                  hipStream_t stream);
size_t bwd_data_lds_bytes(const Geom &g, const BwdDims &bd);   // dynamic LDS of GEMM-1
// most dynamic LDS a GEMM-1 launch may ask for (of the CU's 160 KB): the bound bwd_dims() widens waves_c against,
// mfma_supported() tests, and the launch raises every instance's MaxDynamicSharedMemorySize to, once
constexpr size_t kBwdDataLdsCap = 150 * 1024;
int mfma_bwd_data_f32(const Geom &g, const BwdDims &bd, const Tensors &t, const float *wq,
                      float *gcol, float *ga, float *bias_part, int *cnt, int *table,
                      const float *xt, hipStream_t stream);
int csr_zero_f32(const Geom &g, const BwdDims &bd, int *cnt, hipStream_t stream);
int csr_build_f32(const Geom &g, const BwdDims &bd, const Tensors &t, int *cnt, int *rowptr,
                  void *entries, hipStream_t stream);
int col2im_f32(const Geom &g, const BwdDims &bd, const Tensors &t, const float *gcol,
               const int *rowptr, const void *entries, float *sums, hipStream_t stream);
// 3-D: scatter lists keyed by sample (mfma_csr3d.hip)
int csr_fill3d_f32(const Geom &g, const BwdDims &bd, const Tensors &t, int *cursor,
                   const int *rowptr, void *entries, hipStream_t stream);
int col2im3d_f32(const Geom &g, const BwdDims &bd, const Tensors &t, const float *gcol,
                 const int *rowptr, const void *entries, float *sums, hipStream_t stream);
size_t col2im3d_sums_bytes(const Geom &g);

}  // namespace mdconv
