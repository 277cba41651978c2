// mfma_bwd_weight_cl.hip -- GEMM-2 (grad_weight) with CHANNELS-LAST gathers (fp32, gfx950).
//
// Same contraction as mfma_bwd_weight.hip (M = output channels, N = input channels of one tap,
// K = pixels, split-K), but the column operand is sampled from the channels-last copy of the
// input, xt[b][q][c] (see mfma_fwd_cl.hip for why: with NCHW every lane of a 3-D gather lands in
// its own cache line; in xt a corner of 64 channels is one 256-byte segment).
//
//   * workgroup tile = (WR * MB * 32) output channels x 64 input channels; a thread owns
//     (pixel kk of the 16-pixel chunk, channel quad cq): 2^ND 16-byte loads per chunk, blended
//     into 4 values of the B slab [16 pixels][64 channels];
//   * the tap table entry of (tap, pixel) holds 2^ND corner byte offsets into xt and 2^ND weights
//     (mask, validity and load gating folded in); it is written by GEMM-1 (mfma_bwd_data.hip);
//   * everything else (packed grad_out fragments, double-buffered slab, split-K partials and the
//     reduction) is shared with the NCHW kernel.
#include "mfma_kernels.hpp"
#include "mfma_tile.hpp"

#include <stdlib.h>

namespace mdconv {

namespace {


// <WR, WC, MB, NBW>: WR x WC waves, each MB x NBW blocks of 32 x 32; WC * NBW * 32 == 64
//   <4, 1, 2, 2>  256 x 64   (the narrow tiles -- 64 x 64, 128 x 64 for C_out <= 64 / 128 -- run the 64-pixel-slab
//                             kernel further down, mfma_bwd_weight_cl64_kernel)
template <int ND, bool PADN, int WR, int WC, int MB, int NBW>
__global__ __launch_bounds__(256, 1) void mfma_bwd_weight_cl_kernel(Geom g, BwdDims bd,
                                                                 const float *__restrict__ xt,
                                                                 const float *__restrict__ ga,
                                                                 const int *__restrict__ table,
                                                                 float *__restrict__ part) {
  static_assert(WR * WC == 4 && WC * NBW * 32 == 64, "four waves, 64 input channels");
  constexpr int NC = 1 << ND;
  constexpr int BK = kBK;
  constexpr int RM = WR * MB * 32, CN = WC * NBW * 32;
  constexpr int NH = CN / 64;   // 64-channel halves a thread gathers for
  constexpr int kPitch = CN + 1;
  __shared__ __attribute__((aligned(16))) float Bs[2 * BK * kPitch];

  // unit = split * (mtiles * K * cblks) + (mtile * K + tap) * cblks + cblk.  The linear block id
  // is remapped so that every XCD (own L2) gets a contiguous run of units: the taps of one pixel
  // split gather overlapping neighbourhoods of xt and read the same grad_out slabs, and now do so
  // behind the same L2 (with tap-major dispatch order every tap streamed xt from HBM on its own:
  // 8.7 GB of L2 misses per launch at cfg4, 6 TB/s -- the kernel was HBM-bound)
  const int per_split = gridDim.x;
  const int unit = xcd_remap(blockIdx.x + per_split * blockIdx.y, per_split * gridDim.y);
  const int split = unit / per_split;
  int id = unit - split * per_split;
  const int cblk = id % bd.cblks; id /= bd.cblks;
  const int tap = id % g.K;
  const int mtile = id / g.K;
  const int c0 = cblk * CN;

  const int tid = threadIdx.x, lane = tid & 63, kh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wcn = wave % WC;
  const int cq = tid & 15, kk = tid >> 4;   // channel quad of the tile, pixel within the chunk

  const int pairs_total = bd.Np / 32;
  const int p_begin = split * bd.pairs_per_split;
  const int p_end = min(p_begin + bd.pairs_per_split, pairs_total);
  const int t_begin = 2 * p_begin, t_end = 2 * p_end;   // chunk range (16 pixels each), even count

  const rsrc_t r_xt = make_rsrc(xt, (size_t)g.B * g.S_i * g.C * sizeof(float));
  const int slab_bytes = bd.mblks * 2 * 64 * 16;
  const rsrc_t r_ga = make_rsrc(ga, (size_t)(bd.Np / 16) * slab_bytes);
  const int entry_bytes = 2 * NC * 4;
  const int dg = c0 / g.Cdg;   // the 64-channel tile lies inside one deformable group
  const rsrc_t r_tab = make_rsrc(table + (size_t)(dg * g.K + tap) * bd.Np * (2 * NC),
                                 (size_t)bd.Np * entry_bytes);
  const int wo = mtile * RM + wr * MB * 32;   // first output channel of this wave
  const int a_voff = ((wo / 32) * 2 * 64 + lane) * 16;
  // rows beyond C_out are padding; with conv groups only the output channels of the groups that
  // own this wave's input channels can receive a gradient (block-diagonal dense product)
  bool m_active = wo < g.O;
  if (g.G > 1) {
    const int cw = c0 + wcn * NBW * 32;
    const int o_lo = (min(cw, g.C - 1) / g.Cg) * g.Og;
    const int o_hi = (min(cw + NBW * 32 - 1, g.C - 1) / g.Cg + 1) * g.Og;
    m_active = m_active && wo < o_hi && wo + MB * 32 > o_lo;
  }
  const int t_voff = kk * entry_bytes;
  const int c_voff = (min(c0, g.C - CN) + cq * 4) * 4;   // C is a multiple of the tile width
  f32x16 acc[MB][NBW];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int n = 0; n < NBW; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;

  struct Tab { int off[NC]; float w[NC]; };
  auto load_tab = [&](Tab &tb, int t) {
    const int soff = t * 16 * entry_bytes;
#pragma unroll
    for (int h = 0; h < NC / 4; ++h) {
      const float4 a = buf_load4(r_tab, t_voff + h * 16, soff);
      tb.off[4 * h + 0] = __float_as_int(a.x); tb.off[4 * h + 1] = __float_as_int(a.y);
      tb.off[4 * h + 2] = __float_as_int(a.z); tb.off[4 * h + 3] = __float_as_int(a.w);
      const float4 b = buf_load4(r_tab, t_voff + NC * 4 + h * 16, soff);
      tb.w[4 * h + 0] = b.x; tb.w[4 * h + 1] = b.y; tb.w[4 * h + 2] = b.z; tb.w[4 * h + 3] = b.w;
    }
  };
  float4 rg[NH][NC];
  auto gather = [&](const Tab &tb) {
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int ci = 0; ci < NC; ++ci) rg[h][ci] = buf_load4(r_xt, tb.off[ci] + c_voff + h * 256, 0);
  };
  auto commit = [&](const Tab &tb, int t, float *Bb) {
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int ci = 0; ci < NC; ++ci) {
        s.x = fmaf(tb.w[ci], rg[h][ci].x, s.x); s.y = fmaf(tb.w[ci], rg[h][ci].y, s.y);
        s.z = fmaf(tb.w[ci], rg[h][ci].z, s.z); s.w = fmaf(tb.w[ci], rg[h][ci].w, s.w);
      }
      if (PADN && t * 16 + kk >= g.N) s = make_float4(0.f, 0.f, 0.f, 0.f);
      float *d = Bb + kk * kPitch + h * 64 + cq * 4;
      d[0] = s.x; d[1] = s.y; d[2] = s.z; d[3] = s.w;
    }
  };
  auto load_a = [&](float4 (&ra)[MB][2], int t) {
    if (!m_active) return;
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int q = 0; q < 2; ++q) ra[i][q] = buf_load4(r_ga, a_voff + (i * 2 + q) * 1024, t * slab_bytes);
  };
  auto mma = [&](const float4 (&ra)[MB][2], const float *Bbuf) {
    if (!m_active) return;
    const float *Bb = Bbuf + wcn * NBW * 32 + (lane & 31) + 4 * kh * kPitch;
    // (Reading all B values of the chunk first instead of one step at a time, right before its MFMAs -- hipcc puts
    // one exposed ds_read2 + wait in front of every group of MB * NBW MFMAs -- measured no gain here or in the
    // forward kernels: the other wave of the SIMD fills those gaps.)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float b[NBW];
#pragma unroll
        for (int n = 0; n < NBW; ++n) b[n] = Bb[(8 * q + s) * kPitch + n * 32];
#pragma unroll
        for (int i = 0; i < MB; ++i) {
          const float a = s == 0 ? ra[i][q].x : (s == 1 ? ra[i][q].y : (s == 2 ? ra[i][q].z : ra[i][q].w));
#pragma unroll
          for (int n = 0; n < NBW; ++n)
            acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[n], acc[i][n], 0, 0, 0);
        }
      }
  };

  if (t_begin < t_end) {
    const int t_last = t_end - 1;
    Tab tabA, tabB;
    float4 ra0[MB][2] = {}, ra1[MB][2] = {};
    load_tab(tabA, t_begin);
    load_tab(tabB, t_begin + 1);
    load_a(ra0, t_begin);
    gather(tabA);
    for (int t = t_begin; t < t_end; t += 2) {
      // ---- even chunk ----
      commit(tabA, t, Bs);
      __syncthreads();
      load_a(ra1, t + 1);                   // A first: vmcnt retires in order (see mfma_fwd.hip)
      gather(tabB);                         // chunk t+1
      load_tab(tabA, min(t + 2, t_last));   // chunk t+2
      __builtin_amdgcn_sched_barrier(0);
      mma(ra0, Bs);
      // ---- odd chunk ----
      commit(tabB, t + 1, Bs + BK * kPitch);
      __syncthreads();
      load_a(ra0, min(t + 2, t_last));
      gather(tabA);                         // chunk t+2 (or a harmless repeat at the end)
      load_tab(tabB, min(t + 3, t_last));
      __builtin_amdgcn_sched_barrier(0);
      mma(ra1, Bs + BK * kPitch);
    }
  }

  // partial tile -> part[split][tap][o][c]   (lanes 0-31 = 32 consecutive channels)
#pragma unroll
  for (int n = 0; n < NBW; ++n) {
    float *dst = part + ((size_t)(split * g.K + tap) * bd.OgpB) * bd.Cp + c0 + (wcn * NBW + n) * 32 + (lane & 31);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = wo + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        dst[(size_t)o * bd.Cp] = acc[mb][n][r];
      }
  }
}


// ---------------------------------------------------------------------------------------------
// 64-pixel slabs for the narrow tiles (round 5).  With C_out <= 128 the kernel above does 8 / 16 MFMAs per
// wave between two workgroup barriers (16-pixel chunks) and sends the gathers of a whole chunk at once right
// behind a barrier: at cfg4 (64 x 64 tile, 8 corners of 256 bytes per pixel and tap) it ran at MfmaUtil 38 %
// with a third of its time in gather issue and 15 % in barriers, while the channels-last FORWARD -- the same
// gathers, the same matrix work per (pixel, tap) -- takes half the time with 64-pixel tiles and one barrier
// per 64 x 64 x 64 block.  This variant takes the forward's pipeline: slabs of 64 pixels (4 sub-chunks of 16),
// ONE barrier per slab, the gathers of slab t + 1 issued one pixel group at a time between the sub-chunk
// MFMAs of slab t, and the tap-table entries of a slab staged global -> registers -> LDS two slabs ahead
// (one coalesced 16-byte load per thread) instead of a dependent global load in front of every gather.
// <MB>: 1 = 64 x 64 tile (C_out <= 64), 2 = 128 x 64 (C_out <= 128); 2 x 2 waves.
template <int ND, bool PADN, int MB>
__global__ __launch_bounds__(256) void mfma_bwd_weight_cl64_kernel(Geom g, BwdDims bd,
                                                                   const float *__restrict__ xt,
                                                                   const float *__restrict__ ga,
                                                                   const int *__restrict__ table,
                                                                   float *__restrict__ part) {
  constexpr int NC = 1 << ND;
  constexpr int SW = 2 * NC;            // dwords of a table entry
  constexpr int SP = 64;                // pixels per slab
  constexpr int kPitch = 65;
  constexpr int RM = 2 * MB * 32;
  constexpr int TPT = (SP * SW + 1023) / 1024;   // 16-byte table pieces per thread and slab (1)
  static_assert(TPT == 1, "one table piece per thread");
  __shared__ __attribute__((aligned(16))) float Bs[2 * SP * kPitch];
  __shared__ __attribute__((aligned(16))) int Ts[2 * SP * SW];

  const int per_split = gridDim.x;
  const int unit = xcd_remap(blockIdx.x + per_split * blockIdx.y, per_split * gridDim.y);
  const int split = unit / per_split;
  int id = unit - split * per_split;
  const int cblk = id % bd.cblks; id /= bd.cblks;
  const int tap = id % g.K;
  const int mtile = id / g.K;
  const int c0 = cblk * 64;

  const int tid = threadIdx.x, lane = tid & 63, kh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wcn = wave & 1;
  const int cq = tid & 15, pg = tid >> 4;   // channel quad of the tile, pixel within a 16-pixel group

  const int pairs_total = bd.Np / 32;
  const int p_begin = split * bd.pairs_per_split;
  const int p_end = min(p_begin + bd.pairs_per_split, pairs_total);
  const int t_begin = 2 * p_begin, t_end = 2 * p_end;   // 16-pixel chunks

  const rsrc_t r_xt = make_rsrc(xt, (size_t)g.B * g.S_i * g.C * sizeof(float));
  const int slab_bytes = bd.mblks * 2 * 64 * 16;        // packed grad_out of one 16-pixel chunk
  const rsrc_t r_ga = make_rsrc(ga, (size_t)(bd.Np / 16) * slab_bytes);
  const int entry_bytes = SW * 4;
  const int dg = c0 / g.Cdg;
  const rsrc_t r_tab = make_rsrc(table + (size_t)(dg * g.K + tap) * bd.Np * SW, (size_t)bd.Np * entry_bytes);
  const int wo = mtile * RM + wr * MB * 32;
  const int a_voff = ((wo / 32) * 2 * 64 + lane) * 16;
  bool m_active = wo < g.O;
  if (g.G > 1) {
    const int cw = c0 + wcn * 32;
    const int o_lo = (min(cw, g.C - 1) / g.Cg) * g.Og;
    const int o_hi = (min(cw + 31, g.C - 1) / g.Cg + 1) * g.Og;
    m_active = m_active && wo < o_hi && wo + MB * 32 > o_lo;
  }
  const int c_voff = (min(c0, g.C - 64) + cq * 4) * 4;

  f32x16 acc[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // table slab of the 64 pixels from chunk t on: one 16-byte piece per thread (2-D: the first 128 threads)
  auto load_tab = [&](int t) {
    const bool on = tid * 16 < SP * entry_bytes;
    return buf_load4(r_tab, on ? tid * 16 : 0x7ffffff0, t * 16 * entry_bytes);
  };
  auto store_tab = [&](const float4 &v, int *Tb) {
    if (tid * 16 < SP * entry_bytes) *reinterpret_cast<float4 *>(Tb + tid * 4) = v;
  };
  struct Px { float4 v[NC]; };
  auto issue_px = [&](Px &px, const int *Tb, int p) {
    const int *sp = Tb + p * SW;
    int co[NC];
#pragma unroll
    for (int h = 0; h < NC / 4; ++h) {
      const int4 o4 = *reinterpret_cast<const int4 *>(sp + 4 * h);
      co[4 * h + 0] = o4.x; co[4 * h + 1] = o4.y; co[4 * h + 2] = o4.z; co[4 * h + 3] = o4.w;
    }
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) px.v[ci] = buf_load4(r_xt, co[ci] + c_voff, 0);
  };
  auto commit_px = [&](const Px &px, const int *Tb, int p, bool live, float *Bb) {
    const float *sp = reinterpret_cast<const float *>(Tb + p * SW + NC);
    float w[NC];
#pragma unroll
    for (int h = 0; h < NC / 4; ++h) {
      const float4 w4 = *reinterpret_cast<const float4 *>(sp + 4 * h);
      w[4 * h + 0] = w4.x; w[4 * h + 1] = w4.y; w[4 * h + 2] = w4.z; w[4 * h + 3] = w4.w;
    }
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) {
      s4.x = fmaf(w[ci], px.v[ci].x, s4.x); s4.y = fmaf(w[ci], px.v[ci].y, s4.y);
      s4.z = fmaf(w[ci], px.v[ci].z, s4.z); s4.w = fmaf(w[ci], px.v[ci].w, s4.w);
    }
    if (PADN && !live) s4 = make_float4(0.f, 0.f, 0.f, 0.f);   // pixels of the padded tail: entry all zero, row 0 of xt read
    float *d = Bb + p * kPitch + cq * 4;
    d[0] = s4.x; d[1] = s4.y; d[2] = s4.z; d[3] = s4.w;
  };
  auto load_a = [&](float4 (&ra)[MB][2], int t) {
    if (!m_active) return;
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int q = 0; q < 2; ++q) ra[i][q] = buf_load4(r_ga, a_voff + (i * 2 + q) * 1024, t * slab_bytes);
  };
  auto mma = [&](const float4 (&ra)[MB][2], const float *Bsub) {   // Bsub: 16 pixel rows
    if (!m_active) return;
    const float *Bb = Bsub + wcn * 32 + (lane & 31) + 4 * kh * kPitch;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float b = Bb[(8 * q + s) * kPitch];
#pragma unroll
        for (int i = 0; i < MB; ++i) {
          const float a = s == 0 ? ra[i][q].x : (s == 1 ? ra[i][q].y : (s == 2 ? ra[i][q].z : ra[i][q].w));
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
      }
  };

  if (t_begin < t_end) {
    // ---- prologue: tables of slabs 0 and 1, slab 0 gathered into Bs[0] ----
    {
      const float4 tb0 = load_tab(t_begin);
      const float4 tb1 = load_tab(t_begin + 4);
      store_tab(tb0, Ts);
      store_tab(tb1, Ts + SP * SW);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (t_begin + i < t_end) {
        Px px;
        issue_px(px, Ts, pg + 16 * i);
        commit_px(px, Ts, pg + 16 * i, (t_begin + i) * 16 + pg < g.N, Bs);
      }
    }
    float4 ra0[MB][2] = {}, ra1[MB][2] = {};
    load_a(ra0, t_begin);
    __syncthreads();
    int cur = 0;
    for (int t = t_begin; t < t_end; t += 4) {
      const float *Bcur = Bs + cur * SP * kPitch;
      float *Bnxt = Bs + (cur ^ 1) * SP * kPitch;
      const int *Tn = Ts + (cur ^ 1) * SP * SW;          // table of slab t + 4
      const float4 tb2 = load_tab(t + 8);                // ... of slab t + 8 (beyond the table: zeros)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool nxt_on = t + 4 + j < t_end;           // (wave-uniform) pixel group j of the next slab exists
        const int p = pg + 16 * j;
        Px px;
        // A fragments first, gathers second: vmcnt retires in order (mfma_fwd.hip)
        if (j == 0) load_a(ra1, t + 1);
        if (j == 1) load_a(ra0, t + 2);
        if (j == 2) load_a(ra1, t + 3);
        if (j == 3) load_a(ra0, t + 4);
        if (nxt_on) issue_px(px, Tn, p);
        __builtin_amdgcn_sched_barrier(0);
        if (t + j < t_end) {
          if (j & 1) mma(ra1, Bcur + (16 * j) * kPitch);
          else mma(ra0, Bcur + (16 * j) * kPitch);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (nxt_on) commit_px(px, Tn, p, (t + 4 + j) * 16 + pg < g.N, Bnxt);
      }
      // the table of slab t is dead (its gathers and blends ran during the previous slab): slot for slab t + 8
      store_tab(tb2, Ts + cur * SP * SW);
      __syncthreads();
      cur ^= 1;
    }
  }

  // partial tile -> part[split][tap][o][c]   (lanes 0-31 = 32 consecutive channels)
  float *dst = part + ((size_t)(split * g.K + tap) * bd.OgpB) * bd.Cp + c0 + wcn * 32 + (lane & 31);
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = wo + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      dst[(size_t)o * bd.Cp] = acc[mb][r];
    }
}

}  // namespace


// Resident workgroups per CU of the variant that (nd, padn, wtile, coord) selects: hipOccupancy on the very
// instance, so the split-K count of bwd_dims() follows the register allocation instead of a constant that rots.
#define MDCONV_CL_INSTANCE(ND, PADN, CALL)                                                                     \
  do {                                                                                                          \
    if (wtile == 1) { CALL##64(ND, PADN, 1); }                                                                  \
    else if (wtile == 2) { CALL##64(ND, PADN, 2); }                                                             \
    else { CALL(ND, PADN, 4, 1, 2, 2); }                                                                        \
  } while (0)
#define MDCONV_CL_DISPATCH(CALL)                                                                               \
  do {                                                                                                          \
    if (nd == 2) { if (padn) MDCONV_CL_INSTANCE(2, true, CALL); else MDCONV_CL_INSTANCE(2, false, CALL); }      \
    else { if (padn) MDCONV_CL_INSTANCE(3, true, CALL); else MDCONV_CL_INSTANCE(3, false, CALL); }              \
  } while (0)

int mfma_bwd_weight_cl_occupancy(int nd, bool padn, int wtile) {
  static int cache[2][2][4] = {};
  int &slot = cache[nd == 3][padn][wtile < 0 || wtile > 3 ? 3 : wtile];
  if (slot) return slot;
  int n = 0;
#define OCC_CL(ND, PADN, WR, WC, MB, NBW)                                                                      \
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(                                                          \
      &n, reinterpret_cast<const void *>(&mfma_bwd_weight_cl_kernel<ND, PADN, WR, WC, MB, NBW>), 256, 0)
#define OCC_CL64(ND, PADN, MB)                                                                                 \
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(                                                          \
      &n, reinterpret_cast<const void *>(&mfma_bwd_weight_cl64_kernel<ND, PADN, MB>), 256, 0)
  MDCONV_CL_DISPATCH(OCC_CL);
#undef OCC_CL64
#undef OCC_CL
  (void)hipGetLastError();
  if (n <= 0) n = 3;   // no device (host-only tests): the figures of the committed build
  slot = n;
  return n;
}

int mfma_bwd_weight_cl_launch(const Geom &g, const BwdDims &bd, const float *xt, const float *ga,
                              const int *table, float *part, hipStream_t stream) {
  const dim3 grid(bd.mtiles * g.K * bd.cblks, bd.splits);
  const bool padn = bd.Np != g.N;
  const int nd = g.nd, wtile = bd.wtile;
#define LAUNCH_CL(ND, PADN, WR, WC, MB, NBW)                                                                   \
  hipLaunchKernelGGL((mfma_bwd_weight_cl_kernel<ND, PADN, WR, WC, MB, NBW>), grid, dim3(256), 0,               \
                     stream, g, bd, xt, ga, table, part)
#define LAUNCH_CL64(ND, PADN, MB)                                                                              \
  hipLaunchKernelGGL((mfma_bwd_weight_cl64_kernel<ND, PADN, MB>), grid, dim3(256), 0, stream, g, bd, xt, ga,   \
                     table, part)
  MDCONV_CL_DISPATCH(LAUNCH_CL);
#undef LAUNCH_CL64
#undef LAUNCH_CL
  return check_launch("mfma_bwd_weight_cl");
}

#undef MDCONV_CL_DISPATCH
#undef MDCONV_CL_INSTANCE

}  // namespace mdconv
