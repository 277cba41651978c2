// mfma_fwd_cl.hip -- forward implicit GEMM with CHANNELS-LAST gathers (fp32, gfx950).
//
// Same contraction and tiling idea as mfma_fwd.hip, but the column operand is sampled from a
// channels-last copy of the input, xt[b][q][c] (made once per call by `nchw_to_nhwc`):
//
//   * with NCHW input every lane of a deformable gather lands in its own cache line as soon as
//     the offsets are data dependent -- about 14 lines per 64-lane load in 2-D and 40-64 in 3-D
//     (every lane has its own (dh, dw) row), which is what bounds the 3-D kernels;
//   * in xt one corner of 64 channels is ONE 256-byte segment: a thread fetches 4 consecutive
//     channels of a corner with a 16-byte load, 16 threads cover the corner, a 64-lane load
//     touches 4 corners = 8 lines.  Per sample that is the same number of load instructions as
//     the paired NCHW loads in 3-D (and half in 2-D), at a fraction of the lines.
//
// K slab = 64 channels of one tap (4 MFMA sub-chunks of 16, the packed weights keep their
// 16-channel chunking), so there is one barrier per 64 channels instead of one per 16.
// The sampling state of (tap, pixel) -- 2^ND corner byte offsets into xt and 2^ND weights with
// validity and mask folded in -- is built by BN threads two taps ahead and parked in LDS; the
// gathers of slab s+1 are issued pixel by pixel between the MFMA sub-chunks of slab s.
//
// Used for 3-D shapes with C_in/groups a multiple of 64 and one deformable group (where the
// gathers dominate: DESIGN.md section 4); MDCONV_FWD_CL=0/1 forces it off / on for 2-D too.
#include "mfma_kernels.hpp"
#include "mfma_tile.hpp"

#include <stdio.h>
#include <stdlib.h>

namespace mdconv {

namespace {

constexpr int kSlab = 64;   // channels per LDS slab

// xt[b][q][c] = x[b][c][q]   (32 x 32 tiles through LDS; C is a multiple of 32 here)
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(int C, int S, const float *__restrict__ x,
                                                           float *__restrict__ xt) {
  __shared__ float t[32][33];
  nchw_to_nhwc_tile(t, C, S, x, xt, blockIdx.z, blockIdx.y * 32, blockIdx.x * 32);
}

template <int ND, bool MOD, int BM, int BN, int WM>
__global__ __launch_bounds__(256, 2) void mfma_fwd_cl_kernel(Geom g, PackDims pd,
                                                             const float *__restrict__ xt,
                                                             const float *__restrict__ wp,
                                                             const float *__restrict__ bias,
                                                             const float *__restrict__ offset,
                                                             const float *__restrict__ mask,
                                                             float *__restrict__ output, int ntm,
                                                             int ntn, int full_tiles, int tail_ways, int tail_hi,
                                                             float *__restrict__ part) {
  constexpr int NC = 1 << ND;
  constexpr int MB = WM / 32;
  constexpr int WAVES_M = BM / WM, WAVES_N = 4 / WAVES_M;
  static_assert(WAVES_M * WAVES_N == 4 && BN == 32 * WAVES_N, "tile shape");
  constexpr int PPT = BN / 16;       // pixels per thread per slab (2 or 4)
  constexpr int BNP = BN + 1;        // LDS pitch of a channel row
  constexpr int SW = 2 * NC;         // state words per pixel
  __shared__ __attribute__((aligned(16))) float Bs[2 * kSlab * BNP];
  __shared__ __attribute__((aligned(16))) float St[3 * BN * SW];

  const int grp = blockIdx.y;
  // Blocks [0, full_tiles) own a whole tile; the tiles left over (the last, partly filled dispatch round -- or every tile of a
  // grid smaller than one round) are cut into tap ranges, each block writing its partial tile to `part`
  // (fwd_tail_plan / fwd_tail_reduce_launch, mfma_fwd.hip: the same plan as the NCHW forward).  Round 5: a 3-D forward of a
  // few tiles used to take one whole-tile time (27 taps on an otherwise empty chip) whatever its size.
  int tile, tap_lo = 0, tap_hi = g.K, tail_slot = -1;
  if ((int)blockIdx.x < full_tiles) {
    tile = xcd_remap(blockIdx.x, full_tiles);
  } else {
    tail_slot = blockIdx.x - full_tiles;
    const int hi_slots = tail_hi * (tail_ways + 1);
    int ti, way, wt;
    if (tail_slot < hi_slots) {
      wt = tail_ways + 1;
      ti = tail_slot / wt;
      way = tail_slot - ti * wt;
    } else {
      wt = tail_ways;
      const int r = tail_slot - hi_slots;
      ti = tail_hi + r / wt;
      way = r - (r / wt) * wt;
    }
    tile = full_tiles + ti;
    tap_lo = way * g.K / wt;
    tap_hi = (way + 1) * g.K / wt;
  }
  const int tn = tile / ntm, tm = tile - tn * ntm;
  const int o0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, kh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * 32;
  const int cq = tid & 15, pg = tid >> 4;   // channel quad of the slab, pixel group

  const int spt = g.Cg / kSlab;             // slabs per tap
  const int S = (tap_hi - tap_lo) * spt;
  const int cchunks = pd.Cgp / kBK;
  const int mblks = pd.Ogp / 32;
  const int slab_bytes = mblks * 2 * 64 * 16;   // one 16-channel chunk of packed weights
  const rsrc_t r_xt = make_rsrc(xt, (size_t)g.B * g.S_i * g.C * sizeof(float));
  const rsrc_t r_wp = make_rsrc(wp + (size_t)grp * g.K * cchunks * (slab_bytes / 4),
                                (size_t)g.K * cchunks * slab_bytes);
  const int a_voff = (((o0 + wm0) / 32) * 2 * 64 + lane) * 16;
  const int c_voff = (grp * g.Cg + cq * 4) * 4;   // this thread's channel quad inside a corner row

  // ---- sampling state: thread p < BN owns pixel n0 + p ----
  const int n_s = min(n0 + (tid < BN ? tid : 0), g.N - 1);
  const int b_s = n_s / g.S_o, pix_s = n_s - b_s * g.S_o;
  int oc[ND];
  out_coords<ND>(g, pix_s, oc);
  float dl[ND], ml = 1.f;   // offsets / mask of the tap whose state is built next
  auto fetch_tap = [&](int tap) {
    if (tid < BN) {
      const int64_t ob = ((int64_t)b_s * (ND * g.K) + ND * tap) * g.S_o + pix_s;
#pragma unroll
      for (int a = 0; a < ND; ++a) dl[a] = offset[ob + (int64_t)a * g.S_o];
      if (MOD) ml = mask[((int64_t)b_s * g.K + tap) * g.S_o + pix_s];
    }
  };
  auto build_state = [&](int tap) {   // from dl / ml, into slot tap % 3
    if (tid < BN) {
      int tcd[ND];
      tap_coords<ND>(g, tap, tcd);
      TapCoef<ND, float> tc;
      make_tap<ND, float>(g, oc, tcd, dl, false, tc);
      float *sp = St + ((tap % 3) * BN + tid) * SW;
#pragma unroll
      for (int ci = 0; ci < NC; ++ci) {
        // corners the reference never reads are parked out of the buffer's range (0 * Inf must not happen)
        sp[ci] = __int_as_float(corner_is_read<ND, float>(tc, ci)
                                    ? (b_s * g.S_i + corner_index<ND, float>(tc, ci)) * g.C * 4 : 0x7ffffff0);
        sp[NC + ci] = corner_weight<ND, float>(tc, ci) * ml;
      }
    }
  };

  f32x16 acc[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // ---- gather / blend of ONE pixel of a slab (2^ND 16-byte loads per thread) ----
  struct Px { float4 v[NC]; };
  auto issue_px = [&](Px &px, int tap, int c0, int p) {
    const float *sp = St + ((tap % 3) * BN + p) * SW;
    int co[NC];
#pragma unroll
    for (int h = 0; h < NC / 4; ++h) {
      const float4 o4 = *reinterpret_cast<const float4 *>(sp + 4 * h);
      co[4 * h + 0] = __float_as_int(o4.x); co[4 * h + 1] = __float_as_int(o4.y);
      co[4 * h + 2] = __float_as_int(o4.z); co[4 * h + 3] = __float_as_int(o4.w);
    }
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) px.v[ci] = buf_load4(r_xt, co[ci] + c_voff, c0 * 4);
  };
  auto commit_px = [&](const Px &px, int tap, int p, float *Bb) {
    const float *sp = St + ((tap % 3) * BN + p) * SW + NC;
    float w[NC];
#pragma unroll
    for (int h = 0; h < NC / 4; ++h) {
      const float4 w4 = *reinterpret_cast<const float4 *>(sp + 4 * h);
      w[4 * h + 0] = w4.x; w[4 * h + 1] = w4.y; w[4 * h + 2] = w4.z; w[4 * h + 3] = w4.w;
    }
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) {
      s.x = fmaf(w[ci], px.v[ci].x, s.x); s.y = fmaf(w[ci], px.v[ci].y, s.y);
      s.z = fmaf(w[ci], px.v[ci].z, s.z); s.w = fmaf(w[ci], px.v[ci].w, s.w);
    }
    float *d = Bb + (cq * 4) * BNP + p;
    d[0] = s.x; d[BNP] = s.y; d[2 * BNP] = s.z; d[3 * BNP] = s.w;
  };
  auto load_a = [&](float4 (&ra)[MB][2], int chunk) {
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int q = 0; q < 2; ++q) ra[i][q] = buf_load4(r_wp, a_voff + (i * 2 + q) * 1024, chunk * slab_bytes);
  };
  auto mma = [&](const float4 (&ra)[MB][2], const float *Bsub) {   // Bsub: 16 channel rows
    const float *Bb = Bsub + wn0 + (lane & 31) + 4 * kh * BNP;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float b = Bb[(8 * q + s) * BNP];
#pragma unroll
        for (int i = 0; i < MB; ++i) {
          const float a = s == 0 ? ra[i][q].x : (s == 1 ? ra[i][q].y : (s == 2 ? ra[i][q].z : ra[i][q].w));
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
      }
  };

  // ---- prologue: states of the first two taps of the range, slab 0 into Bs[0] ----
  fetch_tap(tap_lo);
  build_state(tap_lo);
  if (tap_lo + 1 < tap_hi) {
    fetch_tap(tap_lo + 1);
    build_state(tap_lo + 1);
  }
  if (tap_lo + 2 < tap_hi) fetch_tap(tap_lo + 2);   // consumed while the first slab is multiplied
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    Px px;
    issue_px(px, tap_lo, 0, pg + 16 * i);
    commit_px(px, tap_lo, pg + 16 * i, Bs);
  }
  float4 ra0[MB][2], ra1[MB][2];
  load_a(ra0, tap_lo * cchunks);
  __syncthreads();

  int tap = tap_lo, cs = 0;   // tap and slab-in-tap of slab s
  for (int s = 0; s < S; ++s) {
    // slab s+1; after the last slab: a harmless repeat of slab s into the unused buffer
    int tapn = tap, csn = cs + 1;
    if (csn == spt) { csn = 0; ++tapn; }
    const bool last = s + 1 == S;
    const int tap1 = last ? tap : tapn, cs1 = last ? cs : csn;
    const float *Bcur = Bs + (s & 1) * kSlab * BNP;
    float *Bnxt = Bs + ((s + 1) & 1) * kSlab * BNP;
    const int chunk0 = tap * cchunks + cs * 4;       // first 16-channel chunk of slab s
    const int chunk_n = tap1 * cchunks + cs1 * 4;    // ... of slab s+1
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // the pixel of slab s+1 handled around sub-chunk j
      const bool has_px = PPT == 4 || (j & 1) == 0;
      const int p = pg + 16 * (PPT == 4 ? j : j / 2);
      Px px;
      // A fragments first, gathers second: vmcnt retires in order (mfma_fwd.hip)
      if (j == 0) load_a(ra1, chunk0 + 1);
      if (j == 1) load_a(ra0, chunk0 + 2);
      if (j == 2) load_a(ra1, chunk0 + 3);
      if (j == 3) load_a(ra0, chunk_n);
      if (has_px) issue_px(px, tap1, cs1 * kSlab, p);
      __builtin_amdgcn_sched_barrier(0);
      if (j & 1) mma(ra1, Bcur + (16 * j) * BNP);
      else mma(ra0, Bcur + (16 * j) * BNP);
      __builtin_amdgcn_sched_barrier(0);
      if (has_px) commit_px(px, tap1, p, Bnxt);
      if (j == 1 && cs == 0) {
        // first slab of a tap: build the state of tap + 2 (its offsets were requested one tap
        // ago) and request the offsets of tap + 3
        if (tap + 2 < tap_hi) build_state(tap + 2);
        if (tap + 3 < tap_hi) fetch_tap(tap + 3);
      }
    }
    __syncthreads();
    tap = tapn;
    cs = csn;
  }

  if (tail_slot >= 0) {   // partial tile of a tap range: part[tail_slot][o of the tile][pixel of the tile]
    float *dst = part + (size_t)tail_slot * (BM * BN) + wn0 + (lane & 31);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[(wm0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * BN] = acc[mb][r];
    return;
  }
  // ---- epilogue: + bias, store [B, O, S_o] (lanes 0-31 -> 32 consecutive pixels) ----
  const int n_e = n0 + wn0 + (lane & 31);
  if (n_e < g.N) {
    const int b_e = n_e / g.S_o;
    const int pix_e = n_e - b_e * g.S_o;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ol = o0 + wm0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (ol < g.Og) {
          const int och = grp * g.Og + ol;
          const float bv = g.with_bias ? bias[och] : 0.f;
          output[(int64_t)(b_e * g.O + och) * g.S_o + pix_e] = acc[mb][r] + bv;
        }
      }
  }
}

template <int ND, bool MOD, int BM, int BN, int WM>
int launch_cl(const Geom &g, const PackDims &pd, const Tensors &t, const float *wp, const float *xt, float *part,
              hipStream_t stream) {
  const int ntm = (g.Og + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
  // resident slots of this instance (static LDS), clamped like the NCHW forward's: the tail scratch holds one partial
  // tile per slot
  static int slots = 0;
  if (!slots) {
    int n = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(
        &n, reinterpret_cast<const void *>(&mfma_fwd_cl_kernel<ND, MOD, BM, BN, WM>), 256, 0);
    (void)hipGetLastError();
    int cus = 0, dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    slots = (cus > 0 ? cus : 256) * (n > 0 ? (n < kTailMaxPerCu ? n : kTailMaxPerCu) : 2);
  }
  int full_tiles, ways, n_hi;
  fwd_tail_plan(g, ntm * ntn, part ? slots : 0, &full_tiles, &ways, &n_hi);
  {
    // a range must keep enough 64-channel slabs to pay for its prologue, its partial tile and the reduction: with C_in = 64
    // in 2-D a tile is 9 slabs, and halves of it measured 9 % SLOWER than whole tiles (C = 64, 56 x 56, B = 8: 51.5 -> 56.1 us)
    const int most = ways + (n_hi > 0 ? 1 : 0);
    if (most > 1 && (g.K / most) * (g.Cg / kSlab) < 6) {
      full_tiles = ntm * ntn;
      ways = 1;
      n_hi = 0;
    }
  }
  const int tail_tiles = ntm * ntn - full_tiles;
  static const bool debug_plan = getenv("MDCONV_DEBUG_PLAN") != nullptr;
  if (debug_plan)
    fprintf(stderr, "[mdconv] forward plan (channels-last): %d x %d tile, %d tiles, slots %d, full %d, tail %d x %d tap ranges (%d of them x %d)\n",
            BM, BN, ntm * ntn, part ? slots : 0, full_tiles, tail_tiles, ways, n_hi, ways + 1);
  hipLaunchKernelGGL((mfma_fwd_cl_kernel<ND, MOD, BM, BN, WM>), dim3(full_tiles + tail_tiles * ways + n_hi, g.G), dim3(256), 0,
                     stream, g, pd, xt, wp, (const float *)t.bias, (const float *)t.offset,
                     (const float *)t.mask, (float *)t.output, ntm, ntn, full_tiles, ways, n_hi, part);
  const int rc = check_launch("mfma_fwd_cl");
  if (rc || tail_tiles == 0) return rc;
  return fwd_tail_reduce_launch(BM, BN, g, part, (const float *)t.bias, (float *)t.output, ntm, tail_tiles, full_tiles, ways,
                                n_hi, stream);
}

template <int ND, bool MOD>
int launch_cl_tiles(const Geom &g, const PackDims &pd, const Tensors &t, const float *wp,
                    const float *xt, float *part, hipStream_t stream) {
  if (pd.BM == 256) return launch_cl<ND, MOD, 256, 32, 64>(g, pd, t, wp, xt, part, stream);
  if (pd.BM == 128) return launch_cl<ND, MOD, 128, 64, 64>(g, pd, t, wp, xt, part, stream);
  return launch_cl<ND, MOD, 64, 64, 32>(g, pd, t, wp, xt, part, stream);
}

}  // namespace

bool fwd_channels_last(const Geom &g) {
  if (g.DG != 1 || g.Cg % kSlab) return false;
  static const int env = getenv("MDCONV_FWD_CL") ? atoi(getenv("MDCONV_FWD_CL")) : -1;   // read once
  if (env >= 0) return env != 0;
  // 3-D always; 2-D for the narrow shapes, where the 256-output NCHW tile is mostly padding
  // (C = O = 64, 56x56, B = 32: 0.19 -> 0.14 ms; C = O = 128: 0.38 -> 0.36; wider shapes measure
  // equal or 1-4 % slower with the layout pass, tools/cl_sweep.py)
  return g.nd == 3 || (g.N >= 16384 && g.C <= 128 && g.O <= 128);
}

size_t fwd_cl_bytes(const Geom &g) { return (size_t)g.B * g.S_i * g.C * sizeof(float); }

// GEMM-2 of the backward (mfma_bwd_weight_cl.hip) under the same conditions
bool bwd_channels_last(const Geom &g) {
  if (g.C % kSlab || (g.DG != 1 && g.Cdg % kSlab)) return false;   // a 64-channel block = one group
  static const int env = getenv("MDCONV_BWD_CL") ? atoi(getenv("MDCONV_BWD_CL")) : -1;   // read once
  if (env >= 0) return env != 0;
  // 3-D always.  2-D: the NCHW pair gathers cost ~30 L1 accesses per 512-byte load instruction at
  // cfg2 (random offsets put every lane in its own sector; the three GEMMs keep the L1 70-75 %
  // busy), the channels-last ones 16 per KiB -- with the line-wide drain of GEMM-1 and the XCD-aware
  // unit order of GEMM-2 the backward is 4-25 % faster over the 2-D shapes of tools/cl_sweep.py
  // (cfg2 2.85 -> 2.75 ms) once the layout pass is amortised.  Round 4, with GEMM-2's split-K sized to the resident
  // slots: equal at 6 k output pixels, 1-10 % faster from 9 k up (tools/cl_sweep_small.py: B = 2 ... 16 shards of
  // 64 / 128 / 256 channels), so the threshold moved from 16 k to 8 k pixels -- the B = 4 shard of cfg2 included.
  return g.nd == 3 || g.G >= 8 || g.N >= 8192;
}

int nchw_to_nhwc_f32(const Geom &g, const float *x, float *xt, hipStream_t stream) {
  const dim3 tg((g.S_i + 31) / 32, g.C / 32, g.B);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, tg, dim3(256), 0, stream, g.C, g.S_i, x, xt);
  return check_launch("nchw_to_nhwc");
}

int mfma_forward_cl_f32(const Geom &g, const PackDims &pd, const Tensors &t, const float *wp, float *part,
                        float *xt, hipStream_t stream) {
  const int rc = nchw_to_nhwc_f32(g, (const float *)t.input, xt, stream);
  if (rc) return rc;
  if (g.nd == 2)
    return g.modulated ? launch_cl_tiles<2, true>(g, pd, t, wp, xt, part, stream)
                       : launch_cl_tiles<2, false>(g, pd, t, wp, xt, part, stream);
  return g.modulated ? launch_cl_tiles<3, true>(g, pd, t, wp, xt, part, stream)
                     : launch_cl_tiles<3, false>(g, pd, t, wp, xt, part, stream);
}

}  // namespace mdconv
