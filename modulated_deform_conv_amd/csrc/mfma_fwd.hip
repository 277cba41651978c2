// mfma_fwd.hip -- forward as ONE fused implicit GEMM on the gfx950 matrix cores (fp32).
//
//   out[o, n] = sum_{tap, c} W[o, c, tap] * ( mask[tap, n] * interp(input[c], p(tap, n)) )
//
// M = output channels, N = flattened output pixels (b, pix), K = (tap, c), tap-major.
// The column operand (the reference's [C*K, step*S_o] `columns` buffer, 925 MB at cfg2,
// mdeformable_conv.cu:159) is never written to HBM: each workgroup gathers a BK x BN slab of it
// straight into LDS and feeds v_mfma_f32_32x32x2_f32 (exact fp32, MI355X_MICROARCH.md).
//
// Design rule learned on the hardware (tools/ubench_mfma2.hip): the f32 MFMA executes on the
// SIMD's vector-FMA datapath, so a VALU instruction of ANY co-resident wave steals ~3.5 cycles
// from the matrix pipe (NV=128 VALU per 16 MFMAs: 155 -> 108 TFLOP/s even at 4 waves/SIMD).
// The kernel is therefore built to issue almost no VALU work in the K loop:
//   * every thread owns ONE output pixel for the whole kernel; per (tap, deformable group) it
//     builds the sampling state ONCE: 2^ND corner byte-offsets (image / channel-subset base
//     folded in) and 2^ND corner weights (validity and the mask folded in) -- offset/mask are
//     read K times per tile, not C*K times;
//   * gathers are raw buffer loads `buffer_load_dwordx2 v, voff[pair], rsrc(input), soffset`:
//     the two corners that are neighbours along the contiguous axis come with one 8-byte load
//     (make_pairs), and the per-chunk channel base lives in the SGPR soffset, so a gather
//     costs zero VALU;
//   * the weight operand is pre-packed in MFMA-fragment order (mfma_tile.hpp) and fetched with
//     buffer_load_dwordx4 (lane-constant voffset, chunk base in soffset, fragment index in the
//     immediate): zero VALU, never touches LDS;
//   * interpolation = 2^ND FMAs per sample; the B slab is written to a double-buffered k-major
//     LDS tile (n across lanes, conflict-free) at lane-constant addresses;
//   * the chunk loop is unrolled by two (LDS buffer parity and the register sets are
//     compile-time), one barrier per chunk; gathers are requested two chunks ahead, weight
//     fragments one chunk ahead, always before the MFMAs of the current chunk.
// B fragments are read with ds_read_b32: lanes 0-31 / 32-63 hit two different k rows, each 32
// consecutive dwords -> no bank conflicts.
#include "mfma_kernels.hpp"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mfma_tile.hpp"

namespace mdconv {

namespace {


template <int ND, bool MOD, int BM, int BN, int WM, int WN, bool PADK>
__global__ __launch_bounds__(256) void mfma_fwd_kernel(Geom g, PackDims pd,
                                                       const float *__restrict__ input,
                                                       const float *__restrict__ wp,
                                                       const float *__restrict__ bias,
                                                       const float *__restrict__ offset,
                                                       const float *__restrict__ mask,
                                                       float *__restrict__ output, int ntm, int ntn,
                                                       int full_tiles, int tail_ways, int tail_hi,
                                                       float *__restrict__ part) {
  constexpr int BK = kBK;
  constexpr int NC = 1 << ND;
  constexpr int NP = NC / 2;                  // corner pairs along the contiguous axis
  constexpr int MB = WM / 32, NB = WN / 32;   // 32x32 MFMA blocks per wave
  constexpr int WAVES_N = BN / WN;
  constexpr int KSUBS = 256 / BN;             // k-subsets of the B-slab generation
  constexpr int CPT = BK / KSUBS;             // channels per thread per chunk
  static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
  static_assert(BK == 16 && CPT >= 1 && BN <= 256 && MB * 2 * 1024 <= 4096, "tile shape");

  __shared__ __attribute__((aligned(16))) float Bs[2 * BK * BN];  // [2][BK][BN]

  // ---- tile assignment (XCD-aware: consecutive tiles -> same XCD L2) ----
  // Blocks [0, full_tiles) own a whole tile (a whole number of dispatch rounds); the tiles left over are cut
  // into `tail_ways` tap ranges each, so that the last, partly filled round is made of short workgroups
  // (launch_fwd_tile: cfg2's 3136 tiles over 1024 slots left 64 workgroups running one per CU for a whole
  // tile time).  A tap-range block writes its partial tile to `part`; fwd_tail_reduce_kernel adds them up.
  const int grp = blockIdx.y;
  int tile, tap_lo = 0, tap_hi = g.K, tail_slot = -1;
  if ((int)blockIdx.x < full_tiles) {
    tile = xcd_remap(blockIdx.x, full_tiles);
  } else {
    // the first tail_hi tail tiles are cut into tail_ways + 1 ranges, the others into tail_ways (fwd_tail_plan)
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
  const int o0 = tm * BM;
  const int n0 = tn * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // SGPR (uniform to the compiler)
  const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
  const int kh = lane >> 5;

  // ---- this thread's output pixel for the gather ----
  const int j = tid % BN, ksub = tid / BN;
  const int n_g = min(n0 + j, g.N - 1);
  const int b_g = n_g / g.S_o;
  const int pix_g = n_g - b_g * g.S_o;
  int oc[ND];
  out_coords<ND>(g, pix_g, oc);

  const int cchunks = pd.Cgp / BK;
  const int T = g.K * cchunks;
  const int mblks = pd.Ogp / 32;
  const int slab_bytes = mblks * 2 * 64 * 4 * 4;   // one chunk of packed weights

  // buffer resources (wave-uniform: built from kernel arguments and blockIdx only)
  const rsrc_t r_in = make_rsrc(input, (size_t)g.B * g.C * g.S_i * sizeof(float));
  const rsrc_t r_wp = make_rsrc(wp + (size_t)grp * T * (slab_bytes / 4), (size_t)T * slab_bytes);
  const int a_voff = (((o0 + wm0) / 32) * 2 * 64 + lane) * 16;   // bytes, lane-constant
  // byte offset of (image b_g, first channel of this thread's k-subset); the chunk's channel
  // base is added through the scalar offset
  const int img_voff = (b_g * g.C + grp * g.Cg + ksub * CPT) * g.S_i * 4;

  f32x16 acc[MB][NB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int k = 0; k < NB; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][k][r] = 0.f;

  // Gathers run TWO chunks ahead of the MFMAs (one chunk was not enough to cover the L2/MALL
  // latency: the loads added 0.18 ms to an otherwise 0.97 ms kernel).  voff / wgt describe the
  // (tap, dg) of the chunk being requested; each of the two in-flight register sets remembers
  // the weights it was requested with.
  int voff[NP];       // byte offsets of the corner PAIRS of the chunk being requested
  float wgt[NC];      // their weights * mask: [2*pi] first element, [2*pi+1] second
  float2 rg0[CPT][NP], rg1[CPT][NP];
  float wc0[NC], wc1[NC];
  int cur_tap = -1, cur_dg = -1;
  // Elements of the pairs the reference never reads (a corner outside the image, mdeformable_conv.cu:9-34): their
  // weight is 0, but 0 * Inf = NaN, so a non-finite value in the neighbouring pixel must not be multiplied at all.
  // A pair with NO element to read is parked out of the buffer's range (free); a pair with ONE such element -- the
  // sample sits across the first or the last column -- keeps its load, and the waves that hold such a lane take a
  // select per loaded element in `commit` (lane masks in SGPRs, one v_cndmask each; wave-uniform branch).
  typedef unsigned long long lanemask_t;
  lanemask_t bad[NC], bad0[NC], bad1[NC];   // of the chunk being requested / of the two in-flight sets
#pragma unroll
  for (int ci = 0; ci < NC; ++ci) bad[ci] = bad0[ci] = bad1[ci] = 0ull;

  // request the gathers of chunk (tap, c0) (and rebuild the sampling state when (tap, dg) changes)
  auto issue = [&](float2 (&rg)[CPT][NP], float (&wc)[NC], lanemask_t (&bd)[NC], int tap, int c0) {
    const int dg = g.DG == 1 ? 0 : min(grp * g.Cg + c0, g.C - 1) / g.Cdg;
    if (tap != cur_tap || dg != cur_dg) {
      float delta[ND];
      const int64_t ob = ((int64_t)(b_g * g.DG + dg) * (ND * g.K) + ND * tap) * g.S_o + pix_g;
#pragma unroll
      for (int a = 0; a < ND; ++a) delta[a] = offset[ob + (int64_t)a * g.S_o];
      int tcd[ND];
      tap_coords<ND>(g, tap, tcd);
      TapCoef<ND, float> tc;
      make_tap<ND, float>(g, oc, tcd, delta, false, tc);
      const float m = MOD ? mask[((int64_t)(b_g * g.DG + dg) * g.K + tap) * g.S_o + pix_g] : 1.f;
      int pidx[NP];
      float pwx[NP], pwy[NP];
      make_pairs<ND, float>(g, tc, m, pidx, pwx, pwy);
      bool prx[NP], pry[NP];
      make_pairs_read<ND, float>(g, tc, prx, pry);
#pragma unroll
      for (int pi = 0; pi < NP; ++pi) {
        voff[pi] = (prx[pi] || pry[pi]) ? img_voff + pidx[pi] * 4 : 0x7ffffff0;
        wgt[2 * pi] = pwx[pi];
        wgt[2 * pi + 1] = pwy[pi];
        bad[2 * pi] = __ballot(pry[pi] && !prx[pi]);       // loaded beside a wanted neighbour, not to be used
        bad[2 * pi + 1] = __ballot(prx[pi] && !pry[pi]);
      }
      cur_tap = tap;
      cur_dg = dg;
    }
    const int soff = c0 * g.S_i * 4;   // scalar: channel base of this chunk
#pragma unroll
    for (int i = 0; i < CPT; ++i)
#pragma unroll
      for (int pi = 0; pi < NP; ++pi) {
        rg[i][pi] = buf_load2(r_in, voff[pi], soff + i * g.S_i * 4);
      }
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) { wc[ci] = wgt[ci]; bd[ci] = bad[ci]; }
  };
  // interpolate the gathered corners and publish the B slab of the chunk starting at channel c0
  auto commit = [&](const float2 (&rgl)[CPT][NP], const float (&wc)[NC], const lanemask_t (&bd)[NC], int c0, float *Bb) {
    lanemask_t any_bad = 0ull;
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) any_bad |= bd[ci];
    float2 rg[CPT][NP];
#pragma unroll
    for (int i = 0; i < CPT; ++i)
#pragma unroll
      for (int pi = 0; pi < NP; ++pi) rg[i][pi] = rgl[i][pi];
    if (any_bad != 0ull) {   // wave-uniform (the masks live in SGPRs)
#pragma unroll
      for (int i = 0; i < CPT; ++i)
#pragma unroll
        for (int pi = 0; pi < NP; ++pi) {
          asm("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(rg[i][pi].x) : "v"(rgl[i][pi].x), "s"(bd[2 * pi]));
          asm("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(rg[i][pi].y) : "v"(rgl[i][pi].y), "s"(bd[2 * pi + 1]));
        }
    }
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      float val = wc[0] * rg[i][0].x;
      val = fmaf(wc[1], rg[i][0].y, val);
#pragma unroll
      for (int pi = 1; pi < NP; ++pi) {
        val = fmaf(wc[2 * pi], rg[i][pi].x, val);
        val = fmaf(wc[2 * pi + 1], rg[i][pi].y, val);
      }
      if (PADK) {   // ragged C_in/groups: rows of the padded K range must be exactly zero
        const int cl = c0 + ksub * CPT + i;
        val = cl < g.Cg ? val : 0.f;
      }
      Bb[(ksub * CPT + i) * BN + j] = val;
    }
  };
  auto load_a = [&](float4 (&ra)[MB][2], int soff) {   // soff = chunk index * slab_bytes
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        ra[i][q] = buf_load4(r_wp, a_voff + (i * 2 + q) * 1024, soff);
      }
  };
  auto mma = [&](const float4 (&ra)[MB][2], const float *Bbuf) {
    const float *Bb = Bbuf + wn0 + (lane & 31) + 4 * kh * BN;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float b[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) b[n] = Bb[(8 * q + s) * BN + n * 32];
#pragma unroll
        for (int i = 0; i < MB; ++i) {
          const float a = s == 0 ? ra[i][q].x : (s == 1 ? ra[i][q].y : (s == 2 ? ra[i][q].z : ra[i][q].w));
#pragma unroll
          for (int n = 0; n < NB; ++n)
            acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[n], acc[i][n], 0, 0, 0);
        }
      }
  };

  // The chunk count per tap is even (C_in/groups is padded to 2*BK with zero weights), so the 2x
  // unrolled loop needs no tail, a tap change can only happen after an odd chunk, and every
  // prefetch is unconditional (with an `if (t + 1 < T)` around the requests hipcc lost track of
  // the outstanding-load count and put s_waitcnt vmcnt(0) in front of the MFMAs).  Plain nested
  // counters keep the per-chunk scalar work to a few SALU instructions (a flat chunk index cost
  // ~50 SALU per chunk in integer divisions).
  float4 ra0[MB][2], ra1[MB][2];
  const int a_last = (T - 1) * slab_bytes;
  int a_soff = tap_lo * cchunks * slab_bytes;   // byte offset of the current chunk in the packed weights
  load_a(ra0, a_soff);
  issue(rg0, wc0, bad0, tap_lo, 0);
  issue(rg1, wc1, bad1, tap_lo, BK);
  for (int tap = tap_lo; tap < tap_hi; ++tap) {
    for (int c0 = 0; c0 < pd.Cgp; c0 += 2 * BK) {
      // position of the chunk pair two chunks ahead (past the end: harmless re-request)
      const bool wrap = c0 + 2 * BK >= pd.Cgp;
      const int ntap = wrap ? min(tap + 1, g.K - 1) : tap;
      const int nc0 = wrap ? 0 : c0 + 2 * BK;
      // ---- even chunk: LDS buffer 0, fragments ra0, gathers rg0 ----
      commit(rg0, wc0, bad0, c0, Bs);
      __syncthreads();
      // A first: vmcnt retires in order, so fragments requested AFTER the gathers would make the
      // MFMAs that need them wait for those gathers as well
      load_a(ra1, a_soff + slab_bytes);
      issue(rg0, wc0, bad0, ntap, nc0);
      __builtin_amdgcn_sched_barrier(0);   // keep every request above the MFMA phase
      mma(ra0, Bs);
      // ---- odd chunk: LDS buffer 1, fragments ra1, gathers rg1 ----
      commit(rg1, wc1, bad1, c0 + BK, Bs + BK * BN);
      __syncthreads();
      a_soff += 2 * slab_bytes;
      load_a(ra0, min(a_soff, a_last));
      issue(rg1, wc1, bad1, ntap, nc0 + BK);
      __builtin_amdgcn_sched_barrier(0);
      mma(ra1, Bs + BK * BN);
    }
  }

  if (tail_slot >= 0) {   // partial tile of a tap range: part[tail_slot][o of the tile][pixel of the tile]
    float *dst = part + (size_t)tail_slot * (BM * BN) + wn0 + (lane & 31);
#pragma unroll
    for (int q = 0; q < NB; ++q)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          dst[(wm0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * BN + q * 32] = acc[mb][q][r];
    return;
  }
  // ---- epilogue: + bias, store [B, O, S_o] (lanes 0-31 -> 32 consecutive pixels) ----
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    const int n_e = n0 + wn0 + q * 32 + (lane & 31);
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
            output[(int64_t)(b_e * g.O + och) * g.S_o + pix_e] = acc[mb][q][r] + bv;
          }
        }
    }
  }
}

// output tile of a tail tile = sum of its tap-range partials (+ bias), in a fixed order
constexpr int kTailSlices = 16;
template <int BM, int BN>
__global__ __launch_bounds__(256) void fwd_tail_reduce_kernel(Geom g, const float *__restrict__ part,
                                                              const float *__restrict__ bias,
                                                              float *__restrict__ output, int ntm,
                                                              int full_tiles, int tail_ways, int tail_hi) {
  // grid = (tail tiles, kSlices): a slice of 512 elements per workgroup (one workgroup per tile walked its 8192
  // elements in 32 dependent rounds: 29 us for 64 tiles at cfg2)
  const int tile = full_tiles + blockIdx.x;
  const int tn = tile / ntm, tm = tile - tn * ntm;
  const int ti = blockIdx.x;
  const int slot0 = ti < tail_hi ? ti * (tail_ways + 1) : tail_hi * (tail_ways + 1) + (ti - tail_hi) * tail_ways;
  const int nways = ti < tail_hi ? tail_ways + 1 : tail_ways;
  const float *src = part + (size_t)slot0 * (BM * BN);
  const int e_lo = blockIdx.y * (BM * BN / kTailSlices);
  for (int e = e_lo + threadIdx.x; e < e_lo + BM * BN / kTailSlices; e += 256) {
    const int ol = tm * BM + e / BN, n = tn * BN + e % BN;
    if (ol >= g.Og || n >= g.N) continue;
    float sacc = src[e];
    for (int w = 1; w < nways; ++w) sacc += src[(size_t)w * (BM * BN) + e];
    const int b = n / g.S_o, pix = n - b * g.S_o;
    output[(int64_t)(b * g.O + ol) * g.S_o + pix] = sacc + (g.with_bias ? bias[ol] : 0.f);
  }
}

}  // namespace

// Tail plan of a tile grid (see the kernel): `full` tiles run whole, the others -- the leftover of the last dispatch
// round, or every tile of a grid smaller than one round -- are cut into tap ranges so that the round is FULL: with
// `rem` tiles on `slots` slots, n_hi = slots - rem * w tiles get w + 1 ranges and the rest w = slots / rem (w = 1:
// those stay whole).  Round 4 only split behind a full round and uniformly (cfg2: 64 leftover tiles x 4 ranges); cutting
// EVERY tile of the B = 4 shard in two (784 workgroups on 1024 slots: again a ragged round) had measured slower
// (0.189 -> 0.211 ms).  At most 4 ranges (each pays a prologue, a 32 KB partial tile and its share of the reduction:
// cfg2 1.005 ms unsplit, 0.979 / 0.984 / 0.981 with 2 / 3 / 4 ranges, 1.027 with 9).  MDCONV_FWD_TAIL=0 disables, =1
// restores the round-4 plan (uniform, only behind a full round).
void fwd_tail_plan(const Geom &g, int tiles, int slots, int *full_tiles, int *ways, int *n_hi) {
  static const int tail_env = getenv("MDCONV_FWD_TAIL") ? atoi(getenv("MDCONV_FWD_TAIL")) : 2;
  *full_tiles = tiles;
  *ways = 1;
  *n_hi = 0;
  if (!tail_env || g.G != 1 || g.K < 2 || slots <= 0) return;
  const int rem = tiles % slots;
  if (rem == 0) return;
  int w = slots / rem;
  const int cap = g.K < 4 ? g.K : 4;
  if (tail_env == 1) {
    if (tiles < slots) return;
    if (w > cap) w = cap;
    if (w < 2) return;
    *full_tiles = tiles - rem;
    *ways = w;
    return;
  }
  if (w >= cap) {            // few leftover tiles: the cap, uniformly (the round stays partly empty)
    *full_tiles = tiles - rem;
    *ways = cap;
    return;
  }
  const int hi = slots - rem * w;          // tiles that take one range more: rem * w + hi = slots
  if (w == 1) {                            // the tiles with one range are whole tiles
    if (hi == 0) return;
    *full_tiles = tiles - hi;
    *ways = 1;
    *n_hi = hi;
    return;
  }
  *full_tiles = tiles - rem;
  *ways = w;
  *n_hi = hi;
}

int fwd_tail_reduce_launch(int BM, int BN, const Geom &g, const float *part, const float *bias, float *output, int ntm,
                           int tail_tiles, int full_tiles, int ways, int n_hi, hipStream_t stream) {
  if (tail_tiles <= 0) return MDCONV_OK;
#define FWD_TAIL_RED(M, N)                                                                                        \
  hipLaunchKernelGGL((fwd_tail_reduce_kernel<M, N>), dim3(tail_tiles, kTailSlices), dim3(256), 0, stream, g, part, bias, \
                     output, ntm, full_tiles, ways, n_hi)
  if (BM == 256 && BN == 32) FWD_TAIL_RED(256, 32);
  else if (BM == 128 && BN == 64) FWD_TAIL_RED(128, 64);
  else if (BM == 64 && BN == 128) FWD_TAIL_RED(64, 128);
  else if (BM == 64 && BN == 64) FWD_TAIL_RED(64, 64);
  else { set_error("fwd_tail_reduce: no instance for a %d x %d tile", BM, BN); return MDCONV_ELAUNCH; }
#undef FWD_TAIL_RED
  return check_launch("fwd_tail_reduce");
}

namespace {
template <int ND, bool MOD, int BM, int BN, int WM, int WN, bool PADK>
int fwd_tile_slots() {
  static int slots = 0;
  if (!slots) {
    int n = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(
        &n, reinterpret_cast<const void *>(&mfma_fwd_kernel<ND, MOD, BM, BN, WM, WN, PADK>), 256, 0);
    (void)hipGetLastError();
    // clamped to kTailMaxPerCu: the tail scratch (fwd_tail_bytes) holds one partial tile per slot, and the tail never
    // has more workgroups than slots (advisor, round 4: an instance that got leaner would otherwise overrun it)
    slots = device_cus() * (n > 0 ? (n < kTailMaxPerCu ? n : kTailMaxPerCu) : 3);
  }
  return slots;
}

template <int ND, bool MOD, int BM, int BN, int WM, int WN, bool PADK>
int launch_fwd_tile_k(const Geom &g, const PackDims &pd, const Tensors &t, const float *wp, float *part,
                      hipStream_t stream) {
  const int ntm = (g.Og + BM - 1) / BM;
  const int ntn = (g.N + BN - 1) / BN;
  int full_tiles, ways, n_hi;
  fwd_tail_plan(g, ntm * ntn, part ? fwd_tile_slots<ND, MOD, BM, BN, WM, WN, PADK>() : 0, &full_tiles, &ways, &n_hi);
  const int tail_tiles = ntm * ntn - full_tiles;
  static const bool debug_plan = getenv("MDCONV_DEBUG_PLAN") != nullptr;
  if (debug_plan)
    fprintf(stderr, "[mdconv] forward plan: %d x %d tile, %d tiles, slots %d, full %d, tail %d x %d tap ranges (%d of them x %d)\n", BM, BN,
            ntm * ntn, part ? fwd_tile_slots<ND, MOD, BM, BN, WM, WN, PADK>() : 0, full_tiles, tail_tiles, ways, n_hi, ways + 1);
  dim3 grid(full_tiles + tail_tiles * ways + n_hi, g.G);
  hipLaunchKernelGGL((mfma_fwd_kernel<ND, MOD, BM, BN, WM, WN, PADK>), grid, dim3(256), 0, stream,
                     g, pd, (const float *)t.input, wp, (const float *)t.bias,
                     (const float *)t.offset, (const float *)t.mask, (float *)t.output, ntm, ntn,
                     full_tiles, ways, n_hi, part);
  int rc = check_launch("mfma_fwd");
  if (rc || tail_tiles == 0) return rc;
  return fwd_tail_reduce_launch(BM, BN, g, part, (const float *)t.bias, (float *)t.output, ntm, tail_tiles, full_tiles, ways,
                                n_hi, stream);
}

template <int ND, bool MOD, int BM, int BN, int WM, int WN>
int launch_fwd_tile(const Geom &g, const PackDims &pd, const Tensors &t, const float *wp, float *part,
                    hipStream_t stream) {
  if (g.Cg % (2 * kBK)) return launch_fwd_tile_k<ND, MOD, BM, BN, WM, WN, true>(g, pd, t, wp, part, stream);
  return launch_fwd_tile_k<ND, MOD, BM, BN, WM, WN, false>(g, pd, t, wp, part, stream);
}

template <int ND, bool MOD>
int launch_fwd(const Geom &g, const PackDims &pd, const Tensors &t, const float *wp, float *part,
               hipStream_t stream) {
  if (pd.BM == 256) return launch_fwd_tile<ND, MOD, 256, 32, 64, 32>(g, pd, t, wp, part, stream);
  if (pd.BM == 128) return launch_fwd_tile<ND, MOD, 128, 64, 64, 32>(g, pd, t, wp, part, stream);
  return launch_fwd_tile<ND, MOD, 64, 128, 64, 32>(g, pd, t, wp, part, stream);
}

}  // namespace


// scratch for the tap-range partials of the tail tiles: at most one partial tile (every tile shape is 8192
// floats) per resident workgroup, 5 of them per CU at the very most
size_t fwd_tail_bytes(const Geom &g) {
  return g.G == 1 && g.K >= 2 ? (size_t)device_cus() * kTailMaxPerCu * 8192 * sizeof(float) : 0;
}

int mfma_forward_f32(const Geom &g, const PackDims &pd, const Tensors &t, const float *wp, float *part,
                     hipStream_t stream) {
  if (g.nd == 2)
    return g.modulated ? launch_fwd<2, true>(g, pd, t, wp, part, stream)
                       : launch_fwd<2, false>(g, pd, t, wp, part, stream);
  return g.modulated ? launch_fwd<3, true>(g, pd, t, wp, part, stream)
                     : launch_fwd<3, false>(g, pd, t, wp, part, stream);
}

}  // namespace mdconv
