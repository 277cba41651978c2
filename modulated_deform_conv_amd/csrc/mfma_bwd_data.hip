// mfma_bwd_data.hip -- grad_offset / grad_mask / grad_input for fp32 on the gfx950 matrix cores.
//
// Reference structure (mdeformable_conv.cu:412-435): GEMM-1 grad_col = W^T . grad_out, then one
// thread per SAMPLE doing 7 global atomics.  Measured on MI355X (tools/ubench_scatter*.hip) the
// 925 M float atomics of cfg2 cost 24.6 ms scattered / 2.8 ms perfectly coalesced, and LDS
// ds_add_f32 is no better (4.7 ms) -- against a 2.3 ms MFMA budget for the whole iteration.
// So nothing here uses floating-point atomics:
//
//  1. mfma_bwd_data_kernel  (col2im_coord + GEMM-1, fused; details above the kernel)
//     M = input channels, N = output pixels, K = output channels.  A = W pre-packed in
//     MFMA-fragment order (`pack_wq`, block-diagonal dense for conv groups), B = the grad_out
//     tile, resident in LDS for all taps.  In the accumulator layout a lane owns ONE pixel and
//     32 channels, so grad_offset / grad_mask -- sums over channels of grad_col * d(sample) --
//     are reduced in registers, then across the two half-waves with one shuffle and across the
//     64-channel blocks of a deformable group through LDS, and written once per
//     (group, tap, pixel) by their single owner.  grad_col itself is streamed to the workspace
//     CHANNEL-INNERMOST, [b][tap][pix][c] (16-byte stores: a lane holds 4 consecutive
//     channels), for step 3.  The kernel also emits grad_out in the fragment order GEMM-2 wants
//     and runs the counting pass of step 2.
//  2. inverted scatter map  (count [inside step 1] -> scan -> fill, integer atomics only):
//     for every (image, deformable group, input pixel q) the list of corner PAIRS anchored at q:
//     (tap * S_o + output pixel, weight * mask on q, weight * mask on q + 1).  Depends only on
//     offset / mask.
//  3. col2im_gather_kernel / col2im_gather_grouped_kernel
//     grad_input[b][c][q] (+)= sum over the list of q of wx * grad_col[b][tap][n][c]
//                            + sum over the list of q-1 of wy * grad_col[b][tap][n][c]:
//     a WAVE walks 8 consecutive anchors, lanes = channels (4 each), so every list entry is one
//     wave-uniform scalar read plus one fully coalesced 16 B/lane vector read of all channels --
//     no divergence, no atomics; a 32-pixel tile is transposed through LDS and written to the
//     NCHW grad_input with whole-line accesses.  (A first version with lanes = pixels and
//     per-lane list walks took 3.5 ms at cfg2, one list entry per CORNER 0.43 ms; now 0.25 ms.)
#include "mfma_kernels.hpp"
#include "mfma_tile.hpp"
#include <algorithm>
#include <map>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>

namespace mdconv {

namespace {

int grid_for(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

// W[o][c][tap] -> wq[tap][ochunk][cblk][q][lane][s] = W[ochunk*16 + 8q + 4(lane>>5) + s]
//                                                      [cblk*32 + (lane&31)][tap]   (0 padded)
// With groups the weight is expanded to its dense block-diagonal form (0 where o and c belong to
// different groups); GEMM-1 then only walks the o-chunks that can be non-zero for its channels.
__device__ __forceinline__ void pack_wq_items(const Geom &g, int ochunks, int cblks, const float *__restrict__ w,
                                              float *__restrict__ wq, int64_t first, int64_t step) {
  const int64_t total = (int64_t)g.K * ochunks * cblks * 2 * 64;   // float4 units
  for (int64_t i = first; i < total; i += step) {
    int64_t r = i;
    const int lane = (int)(r & 63); r >>= 6;
    const int q = (int)(r & 1); r >>= 1;
    const int cblk = (int)(r % cblks); r /= cblks;
    const int ochunk = (int)(r % ochunks);
    const int tap = (int)(r / ochunks);
    const int c = cblk * 32 + (lane & 31);
    const int ob = ochunk * 16 + 8 * q + 4 * (lane >> 5);
    float v[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int o = ob + s;
      v[s] = (o < g.O && c < g.C && o / g.Og == c / g.Cg)
                 ? w[((int64_t)o * g.Cg + (c % g.Cg)) * g.K + tap] : 0.f;
    }
    reinterpret_cast<float4 *>(wq)[i] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

__global__ __launch_bounds__(256) void pack_wq_kernel(Geom g, int ochunks, int cblks,
                                                      const float *__restrict__ w,
                                                      float *__restrict__ wq) {
  pack_wq_items(g, ochunks, cblks, w, wq, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256);
}

// pack_wq + counter clearing + channels-last input copy in one launch (roles by block range; see bwd_prep_f32)
__global__ __launch_bounds__(256) void bwd_prep_kernel(Geom g, int ochunks, int cblks, const float *__restrict__ w,
                                                       float *__restrict__ wq, int *__restrict__ cnt, int64_t cnt_n,
                                                       const float *__restrict__ x, float *__restrict__ xt,
                                                       int nb_pack, int nb_zero, int qtiles, int ctiles) {
  __shared__ float t[32][33];
  int bid = blockIdx.x;
  if (bid < nb_pack) {
    pack_wq_items(g, ochunks, cblks, w, wq, (int64_t)bid * 256 + threadIdx.x, (int64_t)nb_pack * 256);
    return;
  }
  bid -= nb_pack;
  if (bid < nb_zero) {
    for (int64_t i = (int64_t)bid * 256 + threadIdx.x; i < cnt_n; i += (int64_t)nb_zero * 256) cnt[i] = 0;
    return;
  }
  bid -= nb_zero;
  const int qx = bid % qtiles, cy = (bid / qtiles) % ctiles, b = bid / (qtiles * ctiles);
  nchw_to_nhwc_tile(t, g.C, g.S_i, x, xt, b, cy * 32, qx * 32);
}

// ---------------------------------------------------------------------------------------------
// 1. GEMM-1 + coordinate gradients + grad_col stream
// ---------------------------------------------------------------------------------------------
// Work decomposition: a UNIT is one (pixel tile, tap); a workgroup walks a contiguous unit range.
// The first n_full workgroups (a whole number of dispatch rounds, 2 workgroups per CU) take two
// tiles = 2 K units each; the units of the leftover tiles are spread evenly over one last round of
// short workgroups.  (With one workgroup per tile, cfg2's 3136 tiles over 512 slots left the
// seventh round 1/8 full: ~12 % of the kernel.  A fully persistent grid -- one long unit range
// per slot -- was measured 25 % SLOWER: every workgroup then runs in phase with every other and
// they all hit the same few L2 channels at the same time; 2 / 3 / 6 tiles per workgroup: 1.28 /
// 1.31 / 1.40 ms vs 1.26 for one; 5 / 3 / 1 taps per workgroup: 1.44 / 1.46 / 2.06 ms.  Round 3, with the
// tile load no longer exposed: 1 / 2 / 3 / 6 tiles per workgroup of the full rounds = 1.092 / 1.074 / 1.077 /
// 1.072 ms -- two it is.)  A tile
// split between workgroups
// needs no atomics: grad_col, grad_offset and grad_mask are all per-tap outputs.
//
// The grad_out tile ([C_out] x BNP pixels, 32 KB at cfg2) does not depend on the tap, so it is
// loaded into LDS once per tile, o-innermost, and the K loops of all taps read their B fragments
// from it with two ds_read_b128 per 16 MFMAs: no global B loads, LDS writes or barriers inside
// the tap loop.
//
// Software pipeline inside every wave: the epilogue of iteration i-1 (corner gathers, grad_col
// stores, the corner sums) is DRAINED while the K loop of iteration i runs -- the accumulators of
// i-1 are parked in a second register set.  (A separate epilogue phase added its full latency to
// the kernel: co-resident waves run in lockstep, "another wave covers it" did not happen.)  The
// drain is cut into NBATCH batches (RB accumulator rows each); batch q is gathered at the start
// of the q-th part of the K loop and consumed at its end.  A fragments are prefetched three
// chunks ahead so that the MFMAs issued right after a batch of gathers only wait for loads OLDER
// than the gathers (vmcnt retires in order).
//
// QPQ > 0: the K loop has exactly QPQ * NBATCH quads (4 chunks each) and is emitted as
// straight-line code; QPQ == 0: any shape, runtime loops.

// CL: the drain gathers from the channels-last input copy xt[b][q][c] (mfma_fwd_cl.hip): a lane
// fetches 4 consecutive channels of ONE corner of its pixel with a 16-byte load -- half the load
// instructions of the paired NCHW loads and, in 3-D, a third of the cache lines.

template <int ND, bool MOD, int WAVES_C, int QPQ, bool CL>
__global__ __launch_bounds__(256, 2) void mfma_bwd_data_kernel(
    Geom g, BwdDims bd, const float *__restrict__ input, const float *__restrict__ gout,
    const float *__restrict__ wq, const float *__restrict__ offset, const float *__restrict__ mask,
    float *__restrict__ gcol, float *__restrict__ grad_offset, float *__restrict__ grad_mask,
    float *__restrict__ ga, float *__restrict__ bias_part, int *__restrict__ cnt,
    int *__restrict__ table, const float *__restrict__ xt, int ntiles, int n_full, int n_tail, int tpw) {
  constexpr int NC = 1 << ND, NP = NC / 2;
  constexpr int MB = 2;
  constexpr int WAVES_P = 4 / WAVES_C;
  constexpr int BNP = 32 * WAVES_P;        // pixels per tile
  constexpr int RB = ND == 2 ? 8 : 4;      // accumulator rows per drain batch
  constexpr int NBATCH = MB * 16 / RB;
  constexpr int HS = NBATCH / 2, CPS = NC / HS;   // CL drain: batches per half tile, corners per batch
  constexpr int kOob = 0x7ffffff0;         // out-of-range buffer offset: loads give 0, stores drop
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int T_o = bd.ochunks;              // multiple of 4
  const int gpitch = T_o * 16 + 4;         // floats per pixel row of the grad_out tile
  float *Gs = smem;                        // [BNP][gpitch]
  float *red = smem + BNP * gpitch;        // [bd.tap_group][nblk][ND + 1][BNP]
  const int kTapGroup = bd.tap_group;      // taps whose grad_offset / grad_mask partials are flushed together

  const int tid = threadIdx.x, lane = tid & 63, kh = lane >> 5;
  // wave id as an SGPR: anything derived from threadIdx is 'divergent' to hipcc, and a divergent
  // buffer soffset is wrapped in a readfirstlane waterfall per load (cdna_hip_programming.md T20)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave / WAVES_P, wp = wave % WAVES_P;

  // unit range of this workgroup: the first n_full workgroups take one whole tile each (complete
  // dispatch rounds), the last n_tail share the units of the remaining tiles evenly.
  // Neighbouring ranges share an XCD, hence an L2.
  int u0, u1;
  if ((int)blockIdx.x < n_full) {
    u0 = xcd_remap(blockIdx.x, n_full) * g.K * tpw;
    u1 = u0 + g.K * tpw;
  } else {
    const int j = xcd_remap(blockIdx.x - n_full, n_tail);
    const int64_t tail_units = (int64_t)(ntiles - n_full * tpw) * g.K;
    u0 = n_full * tpw * g.K + (int)(tail_units * j / n_tail);
    u1 = n_full * tpw * g.K + (int)(tail_units * (j + 1) / n_tail);
  }
  if (u0 >= u1) return;

  const int passes = bd.cblks_q / (2 * WAVES_C);
  const int iters = (u1 - u0) * passes;
  const int nquads = T_o / 4;
  // Deformable groups: every 64-channel block (one wave, one pass) lies inside one group, has its
  // own offsets / mask and its own partial grad_offset / grad_mask ("per block" mode); with one
  // group the corner sums run on over the passes and only the WAVES_C waves are reduced.
  const bool per_block = g.DG > 1;
  const int nblk = per_block ? passes * WAVES_C : WAVES_C;
  const int bpd = per_block ? g.Cdg / 64 : nblk;   // 64-channel blocks per deformable group
  const int frag_bytes = 64 * 16;                      // one [lane][4] fragment
  const int chunk_bytes = bd.cblks_q * 2 * frag_bytes; // one ochunk of wq
  const rsrc_t r_in = make_rsrc(CL ? xt : input, (size_t)g.B * g.C * g.S_i * 4);
  const rsrc_t r_wq = make_rsrc(wq, (size_t)g.K * T_o * chunk_bytes);
  const rsrc_t r_gc = make_rsrc(gcol, (size_t)g.B * g.C * g.K * g.S_o * 4);
  const int a_lane = lane * 16;
  const float *Bb = Gs + (wp * 32 + (lane & 31)) * gpitch + 4 * kh;
  // CL drain (line-wide gathers, see `gather`): per wave a parked-accumulator tile Pk[32][64]
  // ([pixel][channel], 16-byte pieces XOR-swizzled by the pixel so that both the accumulator-layout
  // writes and the gather-layout reads are bank-conflict free without padding), and per pixel a
  // state row St[kStRow]: 2^ND corner byte offsets into xt, the grad_col row offset, 2^ND corner sums
  constexpr int kStRow = 2 * NC + 4;
  float *Pk = red + bd.red_floats + wave * (32 * 64);
  int *St = reinterpret_cast<int *>(red + bd.red_floats + 4 * 32 * 64) + wave * (32 * kStRow);
  const int gl_p = lane >> 2, gl_j = lane & 3;   // gather role: pixel of a half tile, lane of its quad
  auto pk_swz = [](int p) { return ((p & 3) << 2) | ((p >> 2) & 3); };
  if (CL) {
    for (int i = lane; i < 32 * 64; i += 64) Pk[i] = 0.f;
    for (int i = lane; i < 32 * kStRow; i += 64) St[i] = kOob;
  }

  // K range of a pass: with conv groups only the o-chunks of the groups that own this wave's 64
  // channels carry non-zero weights (quads of 4 chunks = 64 output channels)
  auto krange = [&](int pass, int &q_lo, int &q_n) {
    if (g.G == 1) { q_lo = 0; q_n = nquads; return; }
    const int cb = (pass * WAVES_C + wc) * 64;
    const int c_lo = min(cb, g.C - 1), c_hi = min(cb + 63, g.C - 1);
    const int o_lo = (c_lo / g.Cg) * g.Og, o_hi = (c_hi / g.Cg + 1) * g.Og;
    q_lo = o_lo / 64;
    q_n = (o_hi + 63) / 64 - q_lo;
  };
  auto a_base = [&](int tap, int pass, int q_lo) {
    return (tap * T_o + q_lo * 4) * chunk_bytes + ((pass * WAVES_C + wc) * 2) * 2 * frag_bytes;
  };
  // A stream of the running iteration (a_cur, n_cur chunks) and of the following one (a_nxt)
  int a_cur = 0, n_cur = 0, a_nxt = 0, b_cur = 0;
  auto a_off = [&](int t) { return t < n_cur ? a_cur + t * chunk_bytes : a_nxt + (t - n_cur) * chunk_bytes; };
  auto load_a = [&](float4 (&ra)[MB][2], int soff) {
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int q = 0; q < 2; ++q)
        ra[i][q] = buf_load4(r_wq, a_lane + (i * 2 + q) * frag_bytes, soff);
  };
  float4 ra0[MB][2], ra1[MB][2], ra2[MB][2], ra3[MB][2];
  {
    int q_lo, q_n;
    krange(0, q_lo, q_n);
    a_cur = a_base(u0 % g.K, 0, q_lo);
    n_cur = 4;   // the first three chunks of the first iteration
    load_a(ra0, a_off(0));
    load_a(ra1, a_off(1));
    load_a(ra2, a_off(2));
  }

  // ---- the pixel this lane owns in the accumulator layout: of the tile whose K loop runs
  // (`c`) and of the tile whose accumulators are parked (`p`) ----
  struct Pix { int n0, b, pix; bool live; int oc[ND]; };
  auto pix_of_tile = [&](int tile, Pix &px) {
    px.n0 = tile * BNP;
    const int n_raw = px.n0 + wp * 32 + (lane & 31);
    px.live = n_raw < g.N;
    const int n_l = px.live ? n_raw : g.N - 1;
    px.b = n_l / g.S_o;
    px.pix = n_l - px.b * g.S_o;
    out_coords<ND>(g, px.pix, px.oc);
  };
  Pix pc, pp;
  pix_of_tile(u0 / g.K, pc);
  pp = pc;

  // ---- grad_out tile -> LDS ----
  auto load_gout_tile = [&](int tile) {
    const int j = tid % BNP, osub = tid / BNP;
    constexpr int OSTEP = 256 / BNP;
    const int n_t = min(tile * BNP + j, g.N - 1);
    const int b_t = n_t / g.S_o, pix_t = n_t - b_t * g.S_o;
    const bool t_live = tile * BNP + j < g.N;
    const float *src = gout + ((int64_t)b_t * g.O) * g.S_o + pix_t;
    float *dst = Gs + j * gpitch;
    const int Opad = T_o * 16;
#pragma unroll 8
    for (int o = osub; o < Opad; o += OSTEP) {
      const float v = src[(int64_t)min(o, g.O - 1) * g.S_o];
      dst[o] = (t_live && o < g.O) ? v : 0.f;
    }
  };

  // The same tile as eight 16-byte loads per thread (4 consecutive pixels of one output channel each), all in
  // flight at once -- the accumulator registers are free at a tile switch.  The scalar version above goes through
  // four dependent rounds of 4-byte loads behind the draining grad_col stores, exposed between two barriers:
  // 17 % of the kernel's wave-cycles at cfg2 (tools/b1_timing.py).
  constexpr int PGS = BNP / 4;             // pixel quads per tile
  constexpr int OPR = 256 / PGS;           // output channels covered by one round of the 256 threads
  constexpr int kPre = 8;                  // rounds in flight together (a 32 KB tile); further rounds follow
  const bool gout_vec = (g.S_o & 3) == 0 && (reinterpret_cast<uintptr_t>(gout) & 15) == 0;
  auto load_gout_tile_vec = [&](int tile) {
    const int Opad = T_o * 16;
    const int rounds = (Opad + OPR - 1) / OPR;
    const int j4 = (tid % PGS) * 4;
    const bool live4 = tile * BNP + j4 < g.N;   // N is a multiple of 4 here: a quad is live or dead as a whole
    const int n4 = min(tile * BNP + j4, g.N - 4);
    const int b4 = n4 / g.S_o, pix4 = n4 - b4 * g.S_o;
    const float *src = gout + ((int64_t)b4 * g.O) * g.S_o + pix4;
    const int o0 = tid / PGS;
    float *dst = Gs + j4 * gpitch + o0;
    for (int r0 = 0; r0 < rounds; r0 += kPre) {
      float4 v[kPre];
#pragma unroll
      for (int i = 0; i < kPre; ++i)   // clamped address: always a valid 16-byte load
        v[i] = *reinterpret_cast<const float4 *>(src + (int64_t)min((r0 + i) * OPR + o0, g.O - 1) * g.S_o);
#pragma unroll
      for (int i = 0; i < kPre; ++i) {
        const int o = (r0 + i) * OPR + o0;
        if (o < Opad) {
          const bool on = live4 && o < g.O;
          float *d = dst + (r0 + i) * OPR;
          d[0] = on ? v[i].x : 0.f;
          d[gpitch] = on ? v[i].y : 0.f;
          d[2 * gpitch] = on ? v[i].z : 0.f;
          d[3 * gpitch] = on ? v[i].w : 0.f;
        }
      }
    }
  };

  // The workgroup that runs tap 0 of a tile also emits the tile in the A-fragment order of GEMM-2
  // (mfma_bwd_weight.hip): ga[nchunk][mblk][q][lane][s] = grad_out[o = mblk*32 + (lane&31)]
  // [n = nchunk*16 + 8q + 4(lane>>5) + s], 0 outside -- the tile is in LDS anyway, so the separate
  // packing pass over grad_out (0.08 ms at cfg2) is gone.
  auto emit_ga = [&](int tile) {
    constexpr int NCH = BNP / 16;
    const int Opad = T_o * 16;
    const int total = NCH * bd.mblks * 2 * 64;   // float4 units
    for (int i = tid; i < total; i += 256) {
      int r = i;
      const int ln = r & 63; r >>= 6;
      const int q = r & 1; r >>= 1;
      const int mblk = r % bd.mblks, nch = r / bd.mblks;
      const int nchunk = tile * NCH + nch;
      if (nchunk * 16 >= bd.Np) continue;
      const int o = mblk * 32 + (ln & 31);
      const float *src = Gs + (nch * 16 + 8 * q + 4 * (ln >> 5)) * gpitch + o;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (o < Opad) v = make_float4(src[0], src[gpitch], src[2 * gpitch], src[3 * gpitch]);
      reinterpret_cast<float4 *>(ga)[((int64_t)nchunk * bd.mblks + mblk) * 128 + q * 64 + ln] = v;
    }
    // ... and the tile's share of grad_bias: bias_part[tile][o] = sum over the tile's pixels
    // (dead pixels are 0 in LDS); summed over tiles in a fixed order by grad_bias_final_kernel
    if (bias_part != nullptr)
      for (int o = tid; o < g.O; o += 256) {
        float sacc = 0.f;
#pragma unroll 8
        for (int px = 0; px < BNP; ++px) sacc += Gs[px * gpitch + o];
        bias_part[(int64_t)tile * g.O + o] = sacc;
      }
  };

  // sampling state of the tap being drained; before the first drain every gather / store goes
  // out of range and the parked accumulators are 0
  int voff[NP], gc_voff = kOob;
  int voffc[NC];   // CL: byte offsets of the 2^ND corners of the lane's pixel in xt (+ 16 * kh)
  // What finish_tap needs of the sampling state, in factored form (round 5: 13 registers instead of the 2^ND weights
  // and ND x 2^ND derivative weights, 40 in 3-D): per outer axis the low / high weight and derivative factor, and for
  // the last axis the factors of the pair's two elements (make_pairs_f: a clamped side collapses onto one column)
  struct Fac { float ol[ND - 1], oh[ND - 1], osl[ND - 1], osh[ND - 1], xw, yw, xs, ys; } fac;
  float mg = 0.f;
  float S[NC];
  float delta_n[ND], m_n = 1.f;   // raw offset / mask of the unit whose K loop is running
  // NCHW drain: elements of the corner pairs that come along with a wanted neighbour but that the reference never reads
  // (the sample sits across the first / last column, mdeformable_conv.cu:256-267).  Their weight is 0, but 0 * Inf = NaN:
  // the waves that hold such a lane select them away in `consume` (lane masks in SGPRs, wave-uniform branch), as the
  // forward does (mfma_fwd.hip); pairs with nothing to read are parked out of the buffer's range.
  typedef unsigned long long lanemask_t;
  lanemask_t bad[NC];
#pragma unroll
  for (int ci = 0; ci < NC; ++ci) bad[ci] = 0ull;
  f32x16 acc[MB], accp[MB];
#pragma unroll
  for (int pi = 0; pi < NP; ++pi) voff[pi] = kOob;
#pragma unroll
  for (int ci = 0; ci < NC; ++ci) voffc[ci] = kOob;
#pragma unroll
  for (int ci = 0; ci < NC; ++ci) S[ci] = 0.f;
#pragma unroll
  for (int a = 0; a < ND; ++a) delta_n[a] = 0.f;
#pragma unroll
  for (int a = 0; a < ND - 1; ++a) fac.ol[a] = fac.oh[a] = fac.osl[a] = fac.osh[a] = 0.f;
  fac.xw = fac.yw = fac.xs = fac.ys = 0.f;
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) accp[i][r] = 0.f;

  // ---- sampling state of (tapp, parked pixel): corner PAIRS (make_pairs) ----
  // `count`: this wave is the one that also counts the scatter targets of (tapp, pixel) for the
  // inverted scatter map (first pass of the CSR build, csr_pass_kernel<.., false> otherwise):
  // fire-and-forget integer atomics that disappear under the MFMAs.
  auto new_tap_state = [&](int tapp, int dgp, bool count) {
    int tcd[ND];
    tap_coords<ND>(g, tapp, tcd);
    TapCoef<ND, float> tc;
    make_tap<ND, float>(g, pp.oc, tcd, delta_n, true, tc);
    mg = (!g.range_gate || tc.inside) ? m_n : 0.f;
    int pidx[NP];
    float px[NP], py[NP];
    make_pairs<ND, float>(g, tc, 1.f, pidx, px, py);
    bool prx[NP], pry[NP];   // elements the reference reads: the channels-last gathers park the others out of range
    make_pairs_read<ND, float>(g, tc, prx, pry);
    if (ND == 3 && count && kh == 0 && pp.live) {
      // 3-D: one list entry per sample, keyed by its low corner (mfma_csr3d.hip): one atomic
      SampleAnchor<ND> sa;
      sample_anchor<ND>(g, tc, 1.f, sa);
      if (sa.on) atomicAdd(cnt + ((int64_t)pp.b * g.DG + dgp) * bd.S_e + sa.qa, 1);
    }
    if (ND == 2 && count && kh == 0 && pp.live) {
      // scatter targets of this sample = its corner PAIRS with a non-zero scatter weight, keyed by
      // the pair's first element (the "anchor"; the col2im gather walks anchors, see below)
      int aidx[NP];
      float ax[NP], ay[NP];
      make_pairs_f<ND, float>(g, tc, tc.wl, tc.wha, 1.f, aidx, ax, ay);
      int *cseg = cnt + ((int64_t)pp.b * g.DG + dgp) * g.S_i;
#pragma unroll
      for (int pi = 0; pi < NP; ++pi)
        if (ax[pi] != 0.f || ay[pi] != 0.f) atomicAdd(cseg + aidx[pi], 1);
    }
    if (count && kh == 0) {
      // ... and writes the tap-table entry GEMM-2 reads for (dgp, tapp, pixel): byte offsets of
      // the corner pairs (image base folded in) + the 2^ND weights with the mask folded in
      // (layout: mfma_bwd_weight.hip); pixels of the padded tail get an all-zero entry
      const int n = pp.n0 + wp * 32 + lane;
      if (n < bd.Np) {
        int *e = table + ((int64_t)(dgp * g.K + tapp) * bd.Np + n) * (2 * NC);
        int ev[2 * NC];
#pragma unroll
        for (int pi = 0; pi < NP; ++pi) {
          if (bd.cl) {   // channels-last GEMM-2: byte offsets of both corners into xt[b][q][c]
            ev[2 * pi] = pp.live && prx[pi] ? (pp.b * g.S_i + pidx[pi]) * g.C * 4 : kOob;
            ev[2 * pi + 1] = pp.live && pry[pi] ? (pp.b * g.S_i + pidx[pi] + 1) * g.C * 4 : kOob;
          } else {
            // NCHW GEMM-2: byte offset of the pair (4-byte aligned), bit 0 / 1 set = first / second element is loaded
            // but not to be used (mfma_bwd_weight.hip selects it away); nothing to read: parked out of range
            ev[pi] = !pp.live ? 0
                     : ((prx[pi] || pry[pi]) ? ((pp.b * g.C * g.S_i + pidx[pi]) * 4) | (pry[pi] && !prx[pi] ? 1 : 0) | (prx[pi] && !pry[pi] ? 2 : 0)
                                             : kOob);
            ev[NP + pi] = 0;
          }
          ev[NC + 2 * pi] = pp.live ? __float_as_int(px[pi] * m_n) : 0;
          ev[NC + 2 * pi + 1] = pp.live ? __float_as_int(py[pi] * m_n) : 0;
        }
#pragma unroll
        for (int q4 = 0; q4 < 2 * NC; q4 += 4)
          *reinterpret_cast<int4 *>(e + q4) = make_int4(ev[q4], ev[q4 + 1], ev[q4 + 2], ev[q4 + 3]);
      }
    }
#pragma unroll
    for (int pi = 0; pi < NP; ++pi) {
      voff[pi] = (prx[pi] || pry[pi]) ? (pp.b * g.C * g.S_i + pidx[pi] + 4 * kh * g.S_i) * 4 : kOob;
      if (!CL) {
        bad[2 * pi] = __ballot(pry[pi] && !prx[pi]);
        bad[2 * pi + 1] = __ballot(prx[pi] && !pry[pi]);
      }
      voffc[2 * pi] = prx[pi] ? (pp.b * g.S_i + pidx[pi]) * g.C * 4 + 16 * kh : kOob;
      voffc[2 * pi + 1] = pry[pi] ? (pp.b * g.S_i + pidx[pi] + 1) * g.C * 4 + 16 * kh : kOob;
    }
    {
      constexpr int L = ND - 1;
      const int lc = tc.last_lc, hc = tc.last_lc + tc.delta[L];
      const int cl = min(lc, g.in_sz[L] - 2);
      fac.xw = (lc == cl ? tc.wl[L] : 0.f) + (hc == cl ? tc.wh[L] : 0.f);
      fac.yw = (lc == cl + 1 ? tc.wl[L] : 0.f) + (hc == cl + 1 ? tc.wh[L] : 0.f);
      fac.xs = (lc == cl ? tc.sl[L] : 0.f) + (hc == cl ? tc.sh[L] : 0.f);
      fac.ys = (lc == cl + 1 ? tc.sl[L] : 0.f) + (hc == cl + 1 ? tc.sh[L] : 0.f);
#pragma unroll
      for (int a = 0; a < L; ++a) { fac.ol[a] = tc.wl[a]; fac.oh[a] = tc.wh[a]; fac.osl[a] = tc.sl[a]; fac.osh[a] = tc.sh[a]; }
    }
    gc_voff = pp.live ? ((((pp.b * g.K + tapp) * g.S_o + pp.pix) * g.C) + 4 * kh) * 4 : kOob;
    if (CL && kh == 0) {
      int *st = St + lane * kStRow;
#pragma unroll
      for (int c4 = 0; c4 < NC; c4 += 4)
        *reinterpret_cast<int4 *>(st + c4) = make_int4(voffc[c4], voffc[c4 + 1], voffc[c4 + 2], voffc[c4 + 3]);
      st[NC] = gc_voff;
    }
    // S[e] = sum over this lane's channels of grad_col * (element e of the corner pairs).  The
    // corner weights and their derivatives do not depend on the channel, so the drain costs
    // 2^ND FMAs per channel and grad_mask / grad_offset are recovered from S once per tap:
    //   grad_mask += sum_ci w[ci] S[ci],   grad_offset_a += m * sum_ci dw[a][ci] S[ci]
    // with w / dw products of per-axis factors (finish_tap forms them from `fac`).
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) S[ci] = 0.f;
  };

  // ---- drain batch q of the parked accumulators (channels cbase_p ..): gather, then consume ----
  // v: RB rows x 2^ND corner values.  NCHW: [row][pair] float2 (one 8-byte load per pair and
  // row); channels-last: [quad of 4 rows][corner] float4 (one 16-byte load per corner and quad).
  struct Batch { float f[RB * NC]; };
  auto gather = [&](int q, int cbase_p, Batch &v) {
    const int mb = (q * RB) / 16, r0 = (q * RB) % 16;
    if (CL) {
      // Line-wide gathers: batch q covers one half of the wave's 32 pixels (16, all 64 channels) and
      // CPS of their corners; a QUAD of lanes owns a pixel and reads a corner's 256-byte segment of
      // xt as four 64-byte pieces (load k: bytes [64k + 16j, +16) for lane j), so every aligned
      // quad of lanes reads 64 contiguous bytes -- the fast case of the texture path.  (With lane =
      // pixel, the accumulator layout, every lane touched its own cache line: 55 L1 accesses per
      // load instruction at cfg4, the texture path busy 97 % of the kernel.)
      const int pxl = (q / HS) * 16 + gl_p;
      const int *st = St + pxl * kStRow + (q % HS) * CPS;
      const bool full = cbase_p + 64 <= g.C;
      if (full) {   // (wave-uniform: the common case gets immediate offsets and no selects)
#pragma unroll
        for (int c = 0; c < CPS; ++c) {
          const int o = st[c] + 16 * gl_j;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float4 x = buf_load4(r_in, o + 64 * k, cbase_p * 4);
            float *d = v.f + (c * 4 + k) * 4;
            d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
          }
        }
      } else {
#pragma unroll
        for (int c = 0; c < CPS; ++c) {
          const int o = st[c] + 16 * gl_j;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float4 x = buf_load4(r_in, cbase_p + 16 * k + 4 * gl_j < g.C ? o + 64 * k : kOob, cbase_p * 4);
            float *d = v.f + (c * 4 + k) * 4;
            d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
          }
        }
      }
    } else {
#pragma unroll
      for (int rr = 0; rr < RB; ++rr) {
        const int r = r0 + rr;
        const int cu = cbase_p + mb * 32 + (r & 3) + 8 * (r >> 2);   // + 4*kh is in the voffset
        const int cs = min(cu, g.C - 5) * g.S_i * 4;   // cu % 8 < 4, so C-5 is the last valid one
#pragma unroll
        for (int pi = 0; pi < NP; ++pi) {
          const float2 x = buf_load2(r_in, voff[pi], cs);
          v.f[(rr * NP + pi) * 2] = x.x;
          v.f[(rr * NP + pi) * 2 + 1] = x.y;
        }
      }
    }
  };
  auto consume = [&](int q, int cbase_p, const Batch &v) {
    if (CL) {
      const int pxl = (q / HS) * 16 + gl_p, cp = q % HS;
      int *st = St + pxl * kStRow;
      const int swz = pk_swz(pxl);
      const bool full = cbase_p + 64 <= g.C;
      float4 gc[4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        gc[k] = *reinterpret_cast<const float4 *>(Pk + pxl * 64 + 4 * ((4 * k + gl_j) ^ swz));
      if (cp == 0) {   // grad_col row: the quad stores 64 contiguous bytes per instruction
        const int gv = st[NC] + 16 * gl_j;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          buf_store4(r_gc, (full || cbase_p + 16 * k + 4 * gl_j < g.C) ? gv + 64 * k : kOob, cbase_p * 4,
                     gc[k].x, gc[k].y, gc[k].z, gc[k].w);
      }
      float s[CPS];
#pragma unroll
      for (int c = 0; c < CPS; ++c) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float *x = v.f + (c * 4 + k) * 4;
          a = fmaf(gc[k].w, x[3], fmaf(gc[k].z, x[2], fmaf(gc[k].y, x[1], fmaf(gc[k].x, x[0], a))));
        }
        s[c] = a;
      }
      // sum over the quad: DPP operands in asm (hipcc emits mov + mov_dpp + add for the builtin);
      // the s_nops cover the VALU-write -> DPP-read hazard, which is not padded for inline asm
      static_assert(CPS == 2, "two corners per step");
      asm("s_nop 1\n\t"
          "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
          "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
          "s_nop 1\n\t"
          "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
          "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
          "s_nop 1"
          : "+v"(s[0]), "+v"(s[1]));
      if (gl_j == 0) *reinterpret_cast<float2 *>(st + NC + 4 + cp * CPS) = make_float2(s[0], s[1]);
      return;
    }
    const int mb = (q * RB) / 16, r0 = (q * RB) % 16;
    // grad_col[b][tap][pix][c]: rows r0+4g .. r0+4g+3 are 4 consecutive channels.  Dead lanes
    // and padded channels store to an out-of-range offset, which the bounds check drops.
#pragma unroll
    for (int gq = 0; gq < RB / 4; ++gq) {
      const int cu4 = cbase_p + mb * 32 + 8 * ((r0 >> 2) + gq);
      const int vo = cu4 < g.C ? gc_voff : kOob;
      buf_store4(r_gc, vo, cu4 * 4, accp[mb][r0 + 4 * gq], accp[mb][r0 + 4 * gq + 1],
                 accp[mb][r0 + 4 * gq + 2], accp[mb][r0 + 4 * gq + 3]);
    }
    // padded channels have grad_col == 0 exactly (zero weight rows), no predicate needed
    lanemask_t any_bad = 0ull;
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) any_bad |= bad[ci];
    if (any_bad != 0ull) {   // wave-uniform: some lane holds a pair with an element the reference does not read
#pragma unroll
      for (int rr = 0; rr < RB; ++rr) {
        const float gc = accp[mb][r0 + rr];
#pragma unroll
        for (int ci = 0; ci < NC; ++ci) {
          float x;
          asm("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(x) : "v"(v.f[(rr * NP + (ci >> 1)) * 2 + (ci & 1)]), "s"(bad[ci]));
          S[ci] = fmaf(gc, x, S[ci]);
        }
      }
    } else {
#pragma unroll
      for (int rr = 0; rr < RB; ++rr) {
        const float gc = accp[mb][r0 + rr];
#pragma unroll
        for (int ci = 0; ci < NC; ++ci) {
          // element ci of the corner pairs = corner ci (pair ci/2, first / second element)
          const float x = v.f[(rr * NP + (ci >> 1)) * 2 + (ci & 1)];
          S[ci] = fmaf(gc, x, S[ci]);
        }
      }
    }
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) asm volatile("" : "+v"(S[ci]));   // pin the updates here
  };

  // CL: after the last batch, the owner lanes of a pixel pick up its reduced corner sums
  auto collect = [&]() {
    if (CL) {
      const float *sr = reinterpret_cast<const float *>(St + (lane & 31) * kStRow + NC + 4);
#pragma unroll
      for (int c4 = 0; c4 < NC; c4 += 4) {
        const float4 t = *reinterpret_cast<const float4 *>(sr + c4);
        S[c4] += t.x; S[c4 + 1] += t.y; S[c4 + 2] += t.z; S[c4 + 3] += t.w;
      }
    }
  };
  // park the accumulators of the finished K loop: registers (NCHW drain) or the LDS tile (CL)
  auto park = [&]() {
    if (CL) {
      const int pix = lane & 31, swz = pk_swz(pix);
      float *dst = Pk + pix * 64;
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
          *reinterpret_cast<float4 *>(dst + 4 * ((i * 8 + 2 * rq + kh) ^ swz)) =
              make_float4(acc[i][4 * rq], acc[i][4 * rq + 1], acc[i][4 * rq + 2], acc[i][4 * rq + 3]);
    } else {
#pragma unroll
      for (int i = 0; i < MB; ++i) accp[i] = acc[i];
    }
  };

  // ---- tap `tapp` of the parked tile is complete for this wave's channels: reduce, park in
  // LDS; `flush` (end of a tap group / of the tile / of the unit range) writes the group out ----
  int grp_lo = (u0 % g.K) % kTapGroup;   // first slot of the current tap group held in `red`
  auto finish_tap = [&](int tapp, int blk, bool flush_ok, bool last) {
    float goff[ND], gm = 0.f;
    {
      // element ci = 2 pi + e of the pairs has weight outer(pi) * (e ? yw : xw), outer(pi) = product over the outer
      // axes of the high / low factor (bit (ND-2-a) of pi); the derivative along axis a swaps that axis' factor
      constexpr int L = ND - 1;
      float Tw[NP], Ts[NP];
#pragma unroll
      for (int pi = 0; pi < NP; ++pi) {
        Tw[pi] = fmaf(fac.xw, S[2 * pi], fac.yw * S[2 * pi + 1]);
        Ts[pi] = fmaf(fac.xs, S[2 * pi], fac.ys * S[2 * pi + 1]);
      }
#pragma unroll
      for (int a = 0; a < ND; ++a) goff[a] = 0.f;
#pragma unroll
      for (int pi = 0; pi < NP; ++pi) {
        float ow = 1.f;
#pragma unroll
        for (int a = 0; a < L; ++a) ow *= ((pi >> (L - 1 - a)) & 1) ? fac.oh[a] : fac.ol[a];
        gm = fmaf(ow, Tw[pi], gm);
        goff[L] = fmaf(ow, Ts[pi], goff[L]);
#pragma unroll
        for (int a = 0; a < L; ++a) {
          float od = 1.f;
#pragma unroll
          for (int a2 = 0; a2 < L; ++a2) {
            const bool hi = (pi >> (L - 1 - a2)) & 1;
            od *= (a2 == a) ? (hi ? fac.osh[a2] : fac.osl[a2]) : (hi ? fac.oh[a2] : fac.ol[a2]);
          }
          goff[a] = fmaf(od, Tw[pi], goff[a]);
        }
      }
#pragma unroll
      for (int a = 0; a < ND; ++a) goff[a] *= mg;
    }
    // reduce over channels: the two half-waves here, the blocks of a group at the flush
    if (!CL) {   // (CL: both half-waves already hold the sums over all 64 channels)
#pragma unroll
      for (int a = 0; a < ND; ++a) goff[a] += __shfl_xor(goff[a], 32, 64);
      gm += __shfl_xor(gm, 32, 64);
    }
    if (bd.red_floats == 0) {
      // one wave owns all channels of its pixels (C_in <= 64, one deformable group): nothing to
      // reduce across waves or passes, the owner lanes write straight to global memory
      if (kh == 0 && pp.live) {
        const int64_t ob = ((int64_t)pp.b * (ND * g.K) + ND * tapp) * g.S_o + pp.pix;
#pragma unroll
        for (int a = 0; a < ND; ++a) {
          float *dst = grad_offset + ob + (int64_t)a * g.S_o;
          *dst = g.acc_data ? *dst + goff[a] : goff[a];
        }
        if (MOD) {
          float *dst = grad_mask + ((int64_t)pp.b * g.K + tapp) * g.S_o + pp.pix;
          *dst = g.acc_data ? *dst + gm : gm;
        }
      }
      return;
    }
    const int slot = tapp % kTapGroup;
    if (kh == 0) {
      float *rp = red + ((slot * nblk + blk) * (ND + 1)) * BNP + wp * 32 + lane;
#pragma unroll
      for (int a = 0; a < ND; ++a) rp[a * BNP] = goff[a];
      rp[ND * BNP] = gm;
    }
    if (flush_ok && (slot == kTapGroup - 1 || tapp == g.K - 1 || last)) {
      __syncthreads();
      // single owner of every (b, dg, tap, pix): plain read-modify-write (or write, mdconv_set_accumulate)
      const int tap0 = tapp - slot;
      const int per_slot = g.DG * (ND + 1) * BNP;
      const int items = (slot + 1 - grp_lo) * per_slot;
      for (int x = tid; x < items; x += 256) {
        const int jj = x % BNP, a = (x / BNP) % (ND + 1), dgi = (x / (BNP * (ND + 1))) % g.DG;
        const int sl = grp_lo + x / per_slot;
        const int n = pp.n0 + jj;
        if (n < g.N && (MOD || a < ND)) {
          float sum = 0.f;
          for (int y = 0; y < bpd; ++y)
            sum += red[((sl * nblk + dgi * bpd + y) * (ND + 1) + a) * BNP + jj];
          const int b = n / g.S_o, pix = n - b * g.S_o, tp = tap0 + sl;
          const int64_t seg = (int64_t)b * g.DG + dgi;
          float *dst = a < ND ? grad_offset + (seg * (ND * g.K) + ND * tp + a) * g.S_o + pix
                              : grad_mask + (seg * g.K + tp) * g.S_o + pix;
          *dst = g.acc_data ? *dst + sum : sum;
        }
      }
      __syncthreads();
      grp_lo = 0;   // the next group starts at its first slot (a new tile starts at tap 0)
    }
  };

  // `fresh`: first chunk of an iteration in the straight-line K loop -- its first MFMAs take a
  // literal zero as C (an inline constant of the instruction), which saves zeroing 32 registers
  auto mma = [&](const float4 (&ra)[MB][2], const float *bp, bool fresh) {
    const float4 b0 = *reinterpret_cast<const float4 *>(bp);
    const float4 b1 = *reinterpret_cast<const float4 *>(bp + 8);
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float4 bq = q == 0 ? b0 : b1;
        const float b = s == 0 ? bq.x : (s == 1 ? bq.y : (s == 2 ? bq.z : bq.w));
#pragma unroll
        for (int i = 0; i < MB; ++i) {
          const float a = s == 0 ? ra[i][q].x : (s == 1 ? ra[i][q].y : (s == 2 ? ra[i][q].z : ra[i][q].w));
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, (fresh && q == 0 && s == 0) ? zero : acc[i], 0, 0, 0);
        }
      }
  };
  // four chunks t .. t+3 of (tap, pass); A runs three chunks ahead (into the next iteration).
  // (Fetching the B fragments a chunk ahead as well was measured slower: 1.24 -> 1.30 ms.)
  auto quad_head = [&](int t) { load_a(ra3, a_off(t + 3)); };
  auto quad_tail = [&](int t, bool fresh) {
    const float *bq = Bb + b_cur;
    __builtin_amdgcn_sched_barrier(0);
    mma(ra0, bq + t * 16, fresh);
    load_a(ra0, a_off(t + 4));
    __builtin_amdgcn_sched_barrier(0);
    mma(ra1, bq + (t + 1) * 16, false);
    load_a(ra1, a_off(t + 5));
    __builtin_amdgcn_sched_barrier(0);
    mma(ra2, bq + (t + 2) * 16, false);
    load_a(ra2, a_off(t + 6));
    __builtin_amdgcn_sched_barrier(0);
    mma(ra3, bq + (t + 3) * 16, false);
  };
  auto quad = [&](int t, bool fresh) { quad_head(t); quad_tail(t, fresh); };

  int tile_c = -1;   // tile whose grad_out is in LDS
  int tapp = 0, passp = 0;   // (tap, pass) of the parked accumulators
  int tile = u0 / g.K, tap = u0 - tile * g.K, pass = 0;   // of the running K loop
  for (int it = 0; it < iters; ++it) {
    if (it > 0 && (passp == 0 || per_block)) {
      const int blk = passp * WAVES_C + wc;
      if (per_block) new_tap_state(tapp, min(blk * 64, g.C - 1) / g.Cdg, blk % bpd == 0 && blk * 64 < g.C);
      else new_tap_state(tapp, 0, wc == 0);
    }
    if (tile != tile_c) {
      if (tile_c >= 0) __syncthreads();   // every wave is done with the previous tile's K loops
      if (gout_vec) load_gout_tile_vec(tile);
      else load_gout_tile(tile);
      pix_of_tile(tile, pc);
      tile_c = tile;
      __syncthreads();
      if (tap == 0 && pass == 0) emit_ga(tile);
    }
    if (pass == 0 || per_block) {
      const int dg = per_block ? min((pass * WAVES_C + wc) * 64, g.C - 1) / g.Cdg : 0;
      const int64_t seg = (int64_t)pc.b * g.DG + dg;
      const int64_t ob = (seg * (ND * g.K) + ND * tap) * g.S_o + pc.pix;
#pragma unroll
      for (int a = 0; a < ND; ++a) delta_n[a] = offset[ob + (int64_t)a * g.S_o];
      if (MOD) m_n = mask[(seg * g.K + tap) * g.S_o + pc.pix];
    }
    // A / B streams of this iteration and the start of the next one (for the prefetch overrun)
    int nq;
    {
      int q_lo, tn = tap, pn = pass + 1;
      krange(pass, q_lo, nq);
      a_cur = a_base(tap, pass, q_lo);
      n_cur = nq * 4;
      b_cur = q_lo * 64;
      if (pn == passes) { pn = 0; tn = tap + 1 == g.K ? 0 : tap + 1; }
      int q_lo2, nq2;
      krange(pn, q_lo2, nq2);
      a_nxt = a_base(tn, pn, q_lo2);
    }
    const int cbase_p = (passp * WAVES_C + wc) * 64;   // channels of the parked accumulators
    if (QPQ == 0) {
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    }

#pragma unroll
    for (int q = 0; q < NBATCH; ++q) {
      Batch v;
      // straight-line path: the A fragment of chunk t+3 is requested BEFORE the gathers, so only
      // the fragment of chunk t+4 (needed four chunks later) queues behind them
      if (QPQ > 0) quad_head(q * QPQ * 4);
      gather(q, cbase_p, v);
      asm volatile("" ::: "memory");   // IR-level fence: keep batch q's gathers here
      __builtin_amdgcn_sched_barrier(0);
      if (QPQ > 0) {
        quad_tail(q * QPQ * 4, q == 0);
#pragma unroll
        for (int jq = 1; jq < QPQ; ++jq) quad((q * QPQ + jq) * 4, false);
      } else {
        for (int qd = nq * q / NBATCH; qd < nq * (q + 1) / NBATCH; ++qd) quad(qd * 4, false);
      }
      __builtin_amdgcn_sched_barrier(0);
      consume(q, cbase_p, v);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    // the parked tap is complete after its last pass (per block: after every pass); its group is
    // flushed at a group / tile end
    collect();
    if (it > 0 && (passp == passes - 1 || per_block))
      finish_tap(tapp, per_block ? passp * WAVES_C + wc : wc, passp == passes - 1, false);
    park();
    pp = pc;
    tapp = tap;
    passp = pass;
    if (++pass == passes) {
      pass = 0;
      if (++tap == g.K) { tap = 0; ++tile; }
    }
  }
  // ---- drain of the last iteration ----
  {
    if (passp == 0 || per_block) {
      const int blk = passp * WAVES_C + wc;
      if (per_block) new_tap_state(tapp, min(blk * 64, g.C - 1) / g.Cdg, blk % bpd == 0 && blk * 64 < g.C);
      else new_tap_state(tapp, 0, wc == 0);
    }
    const int cbase_p = (passp * WAVES_C + wc) * 64;
    // no K loop left to hide the gathers behind: two batches in flight at a time (the accumulator and
    // A-fragment registers are free now), half as many exposed round trips.  With one tile per workgroup this
    // block is 1 of 10 drains: it measured 9.7 % of the kernel's wave-cycles at cfg2 one batch at a time.
    static_assert(NBATCH % 2 == 0, "pairs of drain batches");
#pragma unroll
    for (int q = 0; q < NBATCH; q += 2) {
      Batch v0, v1;
      gather(q, cbase_p, v0);
      gather(q + 1, cbase_p, v1);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      consume(q, cbase_p, v0);
      consume(q + 1, cbase_p, v1);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    collect();
    finish_tap(tapp, per_block ? passp * WAVES_C + wc : wc, true, true);
  }
}


// ---------------------------------------------------------------------------------------------
// 2. inverse scatter map (CSR keyed by (image, deformable group, input pixel))
// ---------------------------------------------------------------------------------------------
// Fill pass of the inverted scatter map.  An ENTRY belongs to the anchor (first element) of a
// corner pair of one sample and carries both scatter weights:
//   (tap * S_o + output pixel, weight on the anchor, weight on anchor + 1, 0)
// -- half as many entries, integer atomics and grad_col row reads as one entry per corner.
template <int ND, bool MOD>
__global__ __launch_bounds__(256) void csr_fill_kernel(Geom g, const float *__restrict__ offset,
                                                       const float *__restrict__ mask,
                                                       int *__restrict__ cursor,
                                                       const int *__restrict__ rowptr,
                                                       int4 *__restrict__ entries) {
  constexpr int NP = 1 << (ND - 1);
  // a segment is one (image, deformable group): offset / mask are laid out [b][dg][...]
  const int64_t total = (int64_t)g.B * g.DG * g.K * g.S_o;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int pix = (int)(i % g.S_o);
    const int tap = (int)((i / g.S_o) % g.K);
    const int seg = (int)(i / g.S_o / g.K);   // b * DG + dg
    int oc[ND], tcd[ND];
    out_coords<ND>(g, pix, oc);
    tap_coords<ND>(g, tap, tcd);
    float delta[ND];
    const int64_t ob = ((int64_t)seg * (ND * g.K) + ND * tap) * g.S_o + pix;
#pragma unroll
    for (int a = 0; a < ND; ++a) delta[a] = offset[ob + (int64_t)a * g.S_o];
    TapCoef<ND, float> tc;
    make_tap<ND, float>(g, oc, tcd, delta, true, tc);
    const float m = MOD ? mask[((int64_t)seg * g.K + tap) * g.S_o + pix] : 1.f;
    int aidx[NP];
    float ax[NP], ay[NP], ux[NP], uy[NP];
    make_pairs_f<ND, float>(g, tc, tc.wl, tc.wha, 1.f, aidx, ux, uy);   // which pairs exist (as counted)
    make_pairs_f<ND, float>(g, tc, tc.wl, tc.wha, m, aidx, ax, ay);     // their weights, mask folded in
#pragma unroll
    for (int pi = 0; pi < NP; ++pi) {
      if (ux[pi] != 0.f || uy[pi] != 0.f) {
        const int q = aidx[pi];
        // the counters double as cursors, counted DOWN: no clearing pass between scan and fill (the order inside a
        // list is the atomics' arrival order either way)
        const int pos = rowptr[(int64_t)seg * (g.S_i + 1) + q] + atomicSub(cursor + (int64_t)seg * g.S_i + q, 1) - 1;
        entries[(int64_t)seg * ((int64_t)g.K * g.S_o * NP) + pos] =
            make_int4(tap * g.S_o + pix, __float_as_int(ax[pi]), __float_as_int(ay[pi]), 0);
      }
    }
  }
}

// counters are cleared by a kernel, not hipMemsetAsync (keeps the whole backward a plain sequence of
// kernel nodes under HIP graph capture)
__global__ __launch_bounds__(256) void zero_int_kernel(int *__restrict__ p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0;
}

// exclusive scan of cnt[seg][0..S) -> rowptr[seg][0..S]: grid (chunks of kScanChunk elements,
// segments).  A workgroup first sums everything before its chunk (the counters are L2-resident and
// that is at most S reads), then scans its chunk -- one workgroup per SEGMENT, as in round 1, left
// 8 workgroups walking 35 k anchors each at the 3-D shards (0.12 / 0.22 ms at cfg4 / cfg5).
__global__ __launch_bounds__(256) void csr_scan_kernel(int S, const int *__restrict__ cnt,
                                            int *__restrict__ rowptr) {
  csr_scan_chunk(S, cnt, rowptr);   // mdconv_common.hpp
}

// ---------------------------------------------------------------------------------------------
// 3. grad_input[b][c][q] (+)= sum over the entries e of anchor q of wx_e * gcol[b][src_e][c]
//                           + sum over the entries e of anchor q-1 of wy_e * gcol[b][src_e][c]
// workgroup = 32 consecutive q of one image x 256 channels; a wave WALKS a run of 8 consecutive
// anchors with two accumulators per lane: `cur` (target = the anchor) and `nxt` (target = anchor
// + 1); after an anchor, cur is the finished target, nxt becomes cur.  Every grad_col row is
// thus read once per corner PAIR.  The run starts one anchor early to pick up the carry; rows of
// the image need no special case: pairs never straddle a row (make_pairs), so the anchor in the
// last column has an empty list and hands a zero carry to the next row.
// The list of an anchor is fetched by ONE coalesced vector load (lane i <- entry i) and broadcast
// with readlane, so there is no dependent scalar-load chain per entry.
// ---------------------------------------------------------------------------------------------
constexpr int kRun = 8;   // targets per wave run

template <int ND>
__global__ __launch_bounds__(256) void col2im_gather_kernel(Geom g, const float *__restrict__ gcol,
                                                            const int *__restrict__ rowptr,
                                                            const int4 *__restrict__ entries,
                                                            float *__restrict__ grad_input) {
  constexpr int NP = 1 << (ND - 1);
  constexpr int QT = 4 * kRun;
  __shared__ float tile[256 * (QT + 1)];   // [c][q], pitch 33
  const int qtiles = (g.S_i + QT - 1) / QT;
  // keep neighbouring q tiles on ONE XCD: the rows they share are then re-read from its L2
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int b = bid / qtiles;
  const int q0 = (bid - b * qtiles) * QT;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const rsrc_t r_gc = make_rsrc(gcol + (size_t)b * g.K * g.S_o * g.C, (size_t)g.K * g.S_o * g.C * 4);
  for (int cb = blockIdx.y * 256; cb < g.C; cb += gridDim.y * 256) {
    const int seg = b * g.DG + cb / g.Cdg;   // C_dg is a multiple of 256 here (or DG == 1)
    const int *rp = rowptr + (int64_t)seg * (g.S_i + 1);
    const int4 *ent = entries + (int64_t)seg * ((int64_t)g.K * g.S_o * NP);
    const int c4 = cb + lane * 4;
    const int c_voff = (c4 < g.C ? c4 : 0) * 4;   // C % 4 == 0
    const int qs = q0 + wave * kRun;              // first target of this wave's run
    float4 cur = make_float4(0.f, 0.f, 0.f, 0.f), nxt = cur;
    for (int a = qs - 1; a < qs + kRun; ++a) {
      if (a >= 0 && a < g.S_i) {
        const int e0 = __builtin_amdgcn_readfirstlane(rp[a]);
        const int e1 = __builtin_amdgcn_readfirstlane(rp[a + 1]);
        for (int base = e0; base < e1; base += 64) {
          const int cnt = min(64, e1 - base);
          const int4 mine = (lane < cnt) ? ent[base + lane] : make_int4(0, 0, 0, 0);
          // 16 independent row loads in flight per step; lanes >= cnt hold weights 0, row 0
          for (int i = 0; i < cnt; i += 16) {
            float4 v[16];
            float wx[16], wy[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
              const int src = __builtin_amdgcn_readlane(mine.x, (i + u) & 63);
              wx[u] = __int_as_float(__builtin_amdgcn_readlane(mine.y, (i + u) & 63));
              wy[u] = __int_as_float(__builtin_amdgcn_readlane(mine.z, (i + u) & 63));
              v[u] = buf_load4(r_gc, c_voff, src * g.C * 4);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
              cur.x = fmaf(wx[u], v[u].x, cur.x); cur.y = fmaf(wx[u], v[u].y, cur.y);
              cur.z = fmaf(wx[u], v[u].z, cur.z); cur.w = fmaf(wx[u], v[u].w, cur.w);
              nxt.x = fmaf(wy[u], v[u].x, nxt.x); nxt.y = fmaf(wy[u], v[u].y, nxt.y);
              nxt.z = fmaf(wy[u], v[u].z, nxt.z); nxt.w = fmaf(wy[u], v[u].w, nxt.w);
            }
          }
        }
      }
      if (a >= qs) {
        float *tp = tile + (lane * 4) * (QT + 1) + (a - q0);
        tp[0] = cur.x; tp[QT + 1] = cur.y; tp[2 * (QT + 1)] = cur.z; tp[3 * (QT + 1)] = cur.w;
      }
      cur = nxt;
      nxt = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    // transpose out: thread = (channel, 8 consecutive q); a wave covers 8 channels x 32 q
    {
      const int cl = threadIdx.x >> 2, qs8 = (threadIdx.x & 3) * 8;
      for (int cc = cl; cc < 256; cc += 64) {
        const int c = cb + cc;
        if (c < g.C) {
          float *dst = grad_input + ((int64_t)b * g.C + c) * g.S_i + q0 + qs8;
          const float *src = tile + cc * (QT + 1) + qs8;
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if (q0 + qs8 + k < g.S_i) dst[k] = g.acc_data ? dst[k] + src[k] : src[k];
        }
      }
    }
    __syncthreads();
  }
}

// Deformable groups narrower than 256 channels: LPD = C_dg / 4 lanes cover one group, so a wave
// walks 64 / LPD runs of ONE group at a time (lane = (run j, channel quad r)); each lane group
// follows its own lists.  Entries are fetched LPD at a time (lane r <- entry r) and broadcast
// inside the lane group with a width-limited shuffle.
template <int ND, int LPD>
__global__ __launch_bounds__(256) void col2im_gather_grouped_kernel(
    Geom g, const float *__restrict__ gcol, const int *__restrict__ rowptr,
    const int4 *__restrict__ entries, float *__restrict__ grad_input) {
  constexpr int NP = 1 << (ND - 1);
  constexpr int QT = 4 * kRun, NQ = 64 / LPD;   // targets per tile, runs per wave
  constexpr int CDG = LPD * 4;                    // channels per deformable group
  constexpr int DPB = 256 / CDG;                  // groups per 256-channel block
  constexpr int RPT = QT / kRun;                  // runs per tile (4)
  constexpr int UB = LPD < 16 ? LPD : 16;         // row loads in flight per step
  __shared__ float tile[256 * (QT + 1)];          // [c][q], pitch 33
  const int qtiles = (g.S_i + QT - 1) / QT;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int b = bid / qtiles;
  const int q0 = (bid - b * qtiles) * QT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane / LPD, r = lane % LPD;
  const rsrc_t r_gc = make_rsrc(gcol + (size_t)b * g.K * g.S_o * g.C, (size_t)g.K * g.S_o * g.C * 4);
  for (int cb = 0; cb < g.C; cb += 256) {
    // work items of the block: (group d of DPB) x (RPT / NQ sets of NQ runs)
    for (int item = wave; item < DPB * (RPT / NQ); item += 4) {
      const int d = item / (RPT / NQ), run = (item % (RPT / NQ)) * NQ + j;
      const int c4 = cb + d * CDG + r * 4;
      const int qs = q0 + run * kRun;
      const bool chan_on = c4 < g.C;
      const int seg = b * g.DG + min(c4, g.C - 1) / g.Cdg;
      const int *rp = rowptr + (int64_t)seg * (g.S_i + 1);
      const int4 *ent = entries + (int64_t)seg * ((int64_t)g.K * g.S_o * NP);
      const int c_voff = (chan_on ? c4 : 0) * 4;
      float4 cur = make_float4(0.f, 0.f, 0.f, 0.f), nxt = cur;
      for (int step = 0; step <= kRun; ++step) {
        const int a = qs - 1 + step;
        const bool on = chan_on && a >= 0 && a < g.S_i;
        const int e0 = on ? rp[a] : 0, e1 = on ? rp[a + 1] : 0;
        for (int base = e0; __any(base < e1); base += LPD) {
          const int cnt = max(0, min(LPD, e1 - base));
          const int4 mine = (r < cnt) ? ent[base + r] : make_int4(0, 0, 0, 0);   // weights 0, row 0 beyond
#pragma unroll
          for (int u0 = 0; u0 < LPD; u0 += UB) {
            float4 v[UB];
            float wx[UB], wy[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
              const int src = __shfl(mine.x, u0 + u, LPD);
              wx[u] = __int_as_float(__shfl(mine.y, u0 + u, LPD));
              wy[u] = __int_as_float(__shfl(mine.z, u0 + u, LPD));
              v[u] = buf_load4(r_gc, src * g.C * 4 + c_voff, 0);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
              cur.x = fmaf(wx[u], v[u].x, cur.x); cur.y = fmaf(wx[u], v[u].y, cur.y);
              cur.z = fmaf(wx[u], v[u].z, cur.z); cur.w = fmaf(wx[u], v[u].w, cur.w);
              nxt.x = fmaf(wy[u], v[u].x, nxt.x); nxt.y = fmaf(wy[u], v[u].y, nxt.y);
              nxt.z = fmaf(wy[u], v[u].z, nxt.z); nxt.w = fmaf(wy[u], v[u].w, nxt.w);
            }
          }
        }
        if (step > 0) {
          float *tp = tile + (d * CDG + r * 4) * (QT + 1) + (a - q0);
          tp[0] = cur.x; tp[QT + 1] = cur.y; tp[2 * (QT + 1)] = cur.z; tp[3 * (QT + 1)] = cur.w;
        }
        cur = nxt;
        nxt = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    __syncthreads();
    {
      const int cl = threadIdx.x >> 2, qs8 = (threadIdx.x & 3) * 8;
      for (int cc = cl; cc < 256; cc += 64) {
        const int c = cb + cc;
        if (c < g.C) {
          float *dst = grad_input + ((int64_t)b * g.C + c) * g.S_i + q0 + qs8;
          const float *src = tile + cc * (QT + 1) + qs8;
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if (q0 + qs8 + k < g.S_i) dst[k] = g.acc_data ? dst[k] + src[k] : src[k];
        }
      }
    }
    __syncthreads();
  }
}


// Few channels (C_in = 64 or 128, one deformable group): with lanes = channel quads only C/4 lanes
// of a wave would carry data, so a wave walks 64 / LPD runs side by side (lane = (run j, channel
// quad r)), all of the same image; tile = 4 * (64 / LPD) runs of kRun targets.
template <int ND, int LPD>
__global__ __launch_bounds__(256) void col2im_gather_narrow_kernel(
    Geom g, const float *__restrict__ gcol, const int *__restrict__ rowptr,
    const int4 *__restrict__ entries, float *__restrict__ grad_input) {
  constexpr int NP = 1 << (ND - 1);
  constexpr int NQ = 64 / LPD, RUNS = 4 * NQ, QT = RUNS * kRun;   // runs per wave / tile, targets
  constexpr int CW = LPD * 4;                                      // channels (== C_in)
  constexpr int UB = LPD < 16 ? LPD : 16;
  __shared__ float tile[CW * (QT + 1)];                            // [c][q]
  const int qtiles = (g.S_i + QT - 1) / QT;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int b = bid / qtiles;
  const int q0 = (bid - b * qtiles) * QT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane / LPD, r = lane % LPD;
  const rsrc_t r_gc = make_rsrc(gcol + (size_t)b * g.K * g.S_o * g.C, (size_t)g.K * g.S_o * g.C * 4);
  const int *rp = rowptr + (int64_t)b * (g.S_i + 1);
  const int4 *ent = entries + (int64_t)b * ((int64_t)g.K * g.S_o * NP);
  const int qs = q0 + (wave * NQ + j) * kRun;
  const int c_voff = r * 16;
  float4 cur = make_float4(0.f, 0.f, 0.f, 0.f), nxt = cur;
  for (int step = 0; step <= kRun; ++step) {
    const int a = qs - 1 + step;
    const bool on = a >= 0 && a < g.S_i;
    const int e0 = on ? rp[a] : 0, e1 = on ? rp[a + 1] : 0;
    for (int base = e0; __any(base < e1); base += LPD) {
      const int cnt = max(0, min(LPD, e1 - base));
      const int4 mine = (r < cnt) ? ent[base + r] : make_int4(0, 0, 0, 0);   // weights 0, row 0 beyond
#pragma unroll
      for (int u0 = 0; u0 < LPD; u0 += UB) {
        float4 v[UB];
        float wx[UB], wy[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int src = __shfl(mine.x, u0 + u, LPD);
          wx[u] = __int_as_float(__shfl(mine.y, u0 + u, LPD));
          wy[u] = __int_as_float(__shfl(mine.z, u0 + u, LPD));
          v[u] = buf_load4(r_gc, src * g.C * 4 + c_voff, 0);
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          cur.x = fmaf(wx[u], v[u].x, cur.x); cur.y = fmaf(wx[u], v[u].y, cur.y);
          cur.z = fmaf(wx[u], v[u].z, cur.z); cur.w = fmaf(wx[u], v[u].w, cur.w);
          nxt.x = fmaf(wy[u], v[u].x, nxt.x); nxt.y = fmaf(wy[u], v[u].y, nxt.y);
          nxt.z = fmaf(wy[u], v[u].z, nxt.z); nxt.w = fmaf(wy[u], v[u].w, nxt.w);
        }
      }
    }
    if (step > 0) {
      float *tp = tile + (r * 4) * (QT + 1) + (a - q0);
      tp[0] = cur.x; tp[QT + 1] = cur.y; tp[2 * (QT + 1)] = cur.z; tp[3 * (QT + 1)] = cur.w;
    }
    cur = nxt;
    nxt = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  // transpose out: rows of 32 consecutive q per channel (whole 128-byte lines)
  for (int x = threadIdx.x; x < CW * (QT / 8); x += 256) {
    const int c = x / (QT / 8), qs8 = (x % (QT / 8)) * 8;
    float *dst = grad_input + ((int64_t)b * g.C + c) * g.S_i + q0 + qs8;
    const float *src = tile + c * (QT + 1) + qs8;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (q0 + qs8 + k < g.S_i) dst[k] = g.acc_data ? dst[k] + src[k] : src[k];
  }
}

}  // namespace

int pack_wq_f32(const Geom &g, const BwdDims &bd, const float *weight, float *wq, hipStream_t stream) {
  const int64_t total = (int64_t)g.K * bd.ochunks * bd.cblks_q * 2 * 64;
  hipLaunchKernelGGL(pack_wq_kernel, dim3(grid_for(total)), dim3(256), 0, stream, g, bd.ochunks,
                     bd.cblks_q, weight, wq);
  return check_launch("pack_wq");
}

int bwd_prep_f32(const Geom &g, const BwdDims &bd, const float *weight, float *wq, int *cnt, const float *x,
                 float *xt, hipStream_t stream) {
  const int64_t pack_total = (int64_t)g.K * bd.ochunks * bd.cblks_q * 2 * 64;
  const int64_t cnt_n = (int64_t)g.B * g.DG * bd.S_e;
  int nb_pack = grid_for(pack_total), nb_zero = grid_for(cnt_n);
  if (nb_pack > 512) nb_pack = 512;
  if (nb_zero > 256) nb_zero = 256;
  const int qtiles = (g.S_i + 31) / 32, ctiles = g.C / 32;
  const int64_t nb_t = xt ? (int64_t)qtiles * ctiles * g.B : 0;
  if (nb_t + nb_pack + nb_zero > 0x7fffffff) {   // (not with tensors below 2 GiB; fall back to the three launches)
    int rc = pack_wq_f32(g, bd, weight, wq, stream);
    if (!rc) rc = csr_zero_f32(g, bd, cnt, stream);
    if (!rc && xt) rc = nchw_to_nhwc_f32(g, x, xt, stream);
    return rc;
  }
  hipLaunchKernelGGL(bwd_prep_kernel, dim3((unsigned)(nb_pack + nb_zero + nb_t)), dim3(256), 0, stream, g, bd.ochunks,
                     bd.cblks_q, weight, wq, cnt, cnt_n, x, xt, nb_pack, nb_zero, qtiles, ctiles);
  return check_launch("bwd_prep");
}

int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      n = v;
    else
      n = 256;
  }
  return n;
}

size_t bwd_data_lds_bytes(const Geom &g, const BwdDims &bd) {
  const int bnp = 32 * (4 / bd.waves_c);
  const int passes = bd.cblks_q / (2 * bd.waves_c);
  const size_t red = (size_t)bd.red_floats;
  // channels-last drain: parked accumulators [4 waves][32][64] + state rows [4 waves][32][2^nd * 2 + 4]
  const size_t cl = bd.cl_drain ? (size_t)4 * 32 * 64 + (size_t)4 * 32 * (2 * (1 << g.nd) + 4) : 0;
  return ((size_t)bnp * (bd.ochunks * 16 + 4) + red + cl) * sizeof(float);
}

int mfma_bwd_data_f32(const Geom &g, const BwdDims &bd, const Tensors &t, const float *wq,
                      float *gcol, float *ga, float *bias_part, int *cnt, int *table,
                      const float *xt, hipStream_t stream) {
  // cnt: per-(image, deformable group, input pixel) counters (zeroed by csr_zero_f32), counted
  // by GEMM-1 (CSR pass 1)
#define LAUNCH_BD_(ND, MOD, WC, QPQ, CL)                                                            \
  do {                                                                                          \
    const int bnp = 32 * (4 / WC);                                                              \
    const int ntiles = (g.N + bnp - 1) / bnp;                                                   \
    const size_t lds = bwd_data_lds_bytes(g, bd);                                               \
    /* Per-instance launch state, shared by every host thread that runs backwards (autograd workers,            */ \
    /* tests/test_gpu_concurrency.py): the dynamic-LDS limit is raised ONCE to the cap mfma_supported() enforces */ \
    /* (a per-launch attribute could be lowered by one thread under another's launch), and the resident-workgroup */ \
    /* count is cached per LDS size (it varies with C_out and K for one instance) under a mutex.                  */ \
    static std::once_flag attr_once;                                                            \
    static hipError_t attr_err = hipSuccess;                                                    \
    std::call_once(attr_once, [] {                                                              \
      attr_err = hipFuncSetAttribute((const void *)mfma_bwd_data_kernel<ND, MOD, WC, QPQ, CL>,  \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwdDataLdsCap); \
    });                                                                                         \
    if (attr_err != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(attr_err)); return MDCONV_ELAUNCH; } \
    /* complete dispatch rounds of one-tile workgroups (2 per CU by registers, fewer by LDS),  */ \
    /* then the units of the leftover tiles spread over one more, shorter, round              */ \
    int occ_q = 0;                                                                              \
    {                                                                                           \
      static std::mutex occ_mu;                                                                 \
      static std::map<size_t, int> occ_by_lds;                                                  \
      std::lock_guard<std::mutex> lock(occ_mu);                                                 \
      auto it = occ_by_lds.find(lds);                                                           \
      if (it == occ_by_lds.end()) {                                                             \
        int nq = 0;                                                                             \
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(                                     \
            &nq, (const void *)mfma_bwd_data_kernel<ND, MOD, WC, QPQ, CL>, 256, lds);           \
        (void)hipGetLastError();                                                                \
        it = occ_by_lds.emplace(lds, nq > 0 ? nq : (lds * 2 <= 160 * 1024 ? 2 : 1)).first;      \
      }                                                                                         \
      occ_q = it->second;                                                                       \
    }                                                                                           \
    const int slots = num_cus() * occ_q;                                                        \
    const int tpw = 2;   /* whole tiles per workgroup of the full rounds */                     \
    const int n_full = ntiles / (slots * tpw) * slots;                                          \
    static const bool debug_plan = getenv("MDCONV_DEBUG_PLAN") != nullptr;                      \
    if (debug_plan)                                                                             \
      fprintf(stderr, "[mdconv] GEMM-1 plan: %d tiles, %zu B LDS, %d resident per CU\n", ntiles, lds, occ_q); \
    const int n_tail = (int)std::min<int64_t>((int64_t)(ntiles - n_full * tpw) * g.K, slots);   \
    hipLaunchKernelGGL((mfma_bwd_data_kernel<ND, MOD, WC, QPQ, CL>), dim3(n_full + n_tail), \
                       dim3(256), lds, stream,                                                  \
                       g, bd, (const float *)t.input, (const float *)t.grad_output, wq,         \
                       (const float *)t.offset, (const float *)t.mask, gcol,                    \
                       (float *)t.grad_offset, (float *)t.grad_mask, ga, bias_part, cnt, table, xt,        \
                       ntiles, n_full, n_tail, tpw);                                            \
  } while (0)
/* channels-last drain only where it pays (3-D) */                                                \
#define LAUNCH_BD(ND, MOD, WC, QPQ)                                                             \
  do {                                                                                          \
    if (xt != nullptr && bd.cl_drain) LAUNCH_BD_(ND, MOD, WC, QPQ, true);                       \
    else LAUNCH_BD_(ND, MOD, WC, QPQ, false);                                                   \
  } while (0)
#define LAUNCH_BD2(ND, MOD)                                                                     \
  do {                                                                                          \
    const int nbatch = ND == 2 ? 4 : 8, nquads = bd.ochunks / 4;                                \
    /* straight-line K loop: 2-D only (the 3-D instances spill ~100 registers) */               \
    const int qpq = (ND == 2 && g.G == 1 && nquads % nbatch == 0 && nquads / nbatch <= 2) ? nquads / nbatch : 0; \
    if (bd.waves_c == 4) {                                                                      \
      if (ND == 2 && qpq == 1) LAUNCH_BD(ND, MOD, 4, (ND == 2 ? 1 : 0));                        \
      else if (ND == 2 && qpq == 2) LAUNCH_BD(ND, MOD, 4, (ND == 2 ? 2 : 0));                   \
      else LAUNCH_BD(ND, MOD, 4, 0);                                                            \
    } else if (bd.waves_c == 2) LAUNCH_BD(ND, MOD, 2, 0);                                       \
    else LAUNCH_BD(ND, MOD, 1, 0);                                                              \
  } while (0)
  if (g.nd == 2) { if (g.modulated) LAUNCH_BD2(2, true); else LAUNCH_BD2(2, false); }
  else { if (g.modulated) LAUNCH_BD2(3, true); else LAUNCH_BD2(3, false); }
#undef LAUNCH_BD2
#undef LAUNCH_BD
#undef LAUNCH_BD_
  return check_launch("mfma_bwd_data");
}

int csr_zero_f32(const Geom &g, const BwdDims &bd, int *cnt, hipStream_t stream) {
  const int64_t cnt_n = (int64_t)g.B * g.DG * bd.S_e;
  hipLaunchKernelGGL(zero_int_kernel, dim3(grid_for(cnt_n)), dim3(256), 0, stream, cnt, cnt_n);
  return check_launch("zero_cnt");
}

// second half of the CSR build (the counting pass ran inside GEMM-1): scan -> fill
int csr_build_f32(const Geom &g, const BwdDims &bd, const Tensors &t, int *cnt, int *rowptr,
                  void *entries, hipStream_t stream) {
  const int64_t samples = (int64_t)g.B * g.DG * g.K * g.S_o;
  int rc;
  hipLaunchKernelGGL(csr_scan_kernel, dim3((bd.S_e + kScanChunk - 1) / kScanChunk, g.B * g.DG), dim3(256), 0,
                     stream, bd.S_e, cnt, rowptr);
  if ((rc = check_launch("csr_scan"))) return rc;
  if (bd.sample_keyed) return csr_fill3d_f32(g, bd, t, cnt, rowptr, entries, stream);
#define LAUNCH_CSR(ND, MOD)                                                                     \
  hipLaunchKernelGGL((csr_fill_kernel<ND, MOD>), dim3(grid_for(samples)), dim3(256), 0, stream,  \
                     g, (const float *)t.offset, (const float *)t.mask, cnt, rowptr,            \
                     (int4 *)entries)
  if (g.modulated) LAUNCH_CSR(2, true); else LAUNCH_CSR(2, false);   // 3-D: csr_fill3d_f32 above
#undef LAUNCH_CSR
  return check_launch("csr_fill");
}

int col2im_f32(const Geom &g, const BwdDims &bd, const Tensors &t, const float *gcol,
               const int *rowptr, const void *entries, float *sums, hipStream_t stream) {
  if (bd.sample_keyed) return col2im3d_f32(g, bd, t, gcol, rowptr, entries, sums, stream);
  const int qtiles = (g.S_i + 31) / 32;
  const dim3 grid(g.B * qtiles, 1);
#define LAUNCH_GG(ND, LPD)                                                                      \
  hipLaunchKernelGGL((col2im_gather_grouped_kernel<ND, LPD>), grid, dim3(256), 0, stream, g,    \
                     gcol, rowptr, (const int4 *)entries, (float *)t.grad_input)
#define LAUNCH_NARROW(ND, LPD)                                                                  \
  do {                                                                                          \
    const int qt = 4 * (64 / LPD) * kRun;                                                       \
    hipLaunchKernelGGL((col2im_gather_narrow_kernel<ND, LPD>), dim3(g.B * ((g.S_i + qt - 1) / qt)), \
                       dim3(256), 0, stream, g, gcol, rowptr, (const int4 *)entries,            \
                       (float *)t.grad_input);                                                  \
  } while (0)
  // (2-D only from here: the pair-keyed lists)
  if (g.DG == 1 && g.C == 64) {
    LAUNCH_NARROW(2, 16);
  } else if (g.DG == 1 && g.C == 128) {
    LAUNCH_NARROW(2, 32);
  } else if (g.DG > 1 && g.Cdg == 64) {
    LAUNCH_GG(2, 16);
  } else if (g.DG > 1 && g.Cdg == 128) {
    LAUNCH_GG(2, 32);
  } else {
    hipLaunchKernelGGL((col2im_gather_kernel<2>), grid, dim3(256), 0, stream, g, gcol, rowptr,
                       (const int4 *)entries, (float *)t.grad_input);
  }
#undef LAUNCH_GG
#undef LAUNCH_NARROW
  return check_launch("col2im_gather");
}

}  // namespace mdconv
