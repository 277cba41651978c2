// hp_bwd3_kernel.hpp -- pixel-stationary backward kernel of the native 16-bit path (gfx950): the kernel template and
// its launch / dispatch templates.  Instantiated per (tensor type, rank) in hp_bwd3_{f16,bf16}_{2d,3d}.hip (four
// translation units that compile side by side); the host side is hp_bwd3.hip.
//
// Reference: mdeformable_conv.cu:412-444, 202-318; 3-D mdeformable_conv3d.cu:515-560, 265-395.
// Per (tap, deformable group, pixel)
//     GEMM-1  gc[c] = sum_o W[o, c, tap] grad_out[o, n]                                (once per tap, all channels)
//     S[ci]   = sum_{c in group} gc[c] x[ci][c]  -> grad_mask, grad_offset;   grad_col row -> workspace
//     col[c]  = mask * sum_ci w[ci] x[ci][c]                                   column row   -> workspace
// GEMM-2 (grad_W = grad_out . col^T) is the dense kernel of hp_gemm2.hip over the column rows.
//
// Why a third structure.  hp_bwd2 is tap-stationary: a workgroup owns one tap and walks a pixel
// range, all its waves pass through GEMM-1 -> gather -> GEMM-2 together (2 barriers per 32-pixel
// tile), the GEMM-2 accumulators pin it to one workgroup per CU, and its corner gathers -- which
// MI355X serves at a rate set by the number of loads in flight (the rows come from the L2 / MALL,
// not the L1: N(0,1) offsets scatter neighbouring pixels' corners) -- are in flight only during the
// gather phase of that one workgroup: 101 cycles per wave-load per CU against 28 for the forward
// kernel, which issues the SAME loads (profiles/r02_hp_counters.md).  This kernel takes the forward's
// shape instead: a wave owns 32 pixels for ALL taps, everything between two taps is wave-private (no
// workgroup barrier inside the gather phase), waves of a CU drift apart and overlap one another's
// gather latency, VALU and matrix work, and two workgroups fit a CU.
//   * grad_out of the wave's 32 pixels is read ONCE (not once per tap): staged through LDS and kept
//     as B fragments (K = output channel, N = pixel) in NKS x 4 registers;
//   * W^T[tap] (fragment-packed, cblks x NKS KB) is staged global -> LDS once per workgroup and tap,
//     single-buffered: the next tap's slab is copied piecewise during the gather phase (the matrix
//     phase that reads it is over by then: barrier B1), visible after barrier B2 at the tap's end;
//   * matrix phase: per 32-channel block NKS MFMAs -> 16-bit -> wave-private LDS tile Gc[pixel][c];
//   * gather phase, lane = (pixel, channel octet of ONE deformable group), LPP ADJACENT lanes per pixel (line-wide
//     gathers, tools/ubench_gather16.hip): the grad_col piece goes to the workspace, the 2^ND corner octets are
//     gathered (two iterations in flight), S accumulates with v_dot2c, col with v_fma_mix, S is reduced
//     over the pixel's lanes with DPP adds and parked in the (group, pixel) state row;
//   * lanes = the wave's pixels build the sampling states (offsets / mask one tap ahead, CSR counting atomic) and
//     finish grad_offset / grad_mask from S (single owner, no atomics).
// Shapes: one conv group, Cp in {32, 64, 128, 256}, 1, 2 or 4 deformable groups of 16 ... 256 channels.
//
// Deformable groups (round 6; reference mdeformable_conv.cu:59, :231: `c / (I / DG)` selects the offsets of a channel).
// A (tap, group) pair is a gather UNIT: the matrix phase of a tap covers all channels, then the groups' units follow one
// another, each with LPP = C_dg / 8 lanes per pixel, its own state rows St[group][pixel] and its own channel window of the
// Gc tile and of the workspace rows.  The two half-waves build the states of two GROUPS of the tap at a time (lanes 0-31:
// group 2r, lanes 32-63: group 2r + 1) and keep the per-axis factors of "their" groups in registers for the finish.  With
// one group the half-waves build two TAPS at a time (round 5): at an even tap t lanes 0-31 build (t, pixel) and lanes
// 32-63 build (t + 1, pixel), whose row they hold in registers (`held`) until tap t + 1 starts.
// Before, shapes with deformable groups ran on hp_bwd2 (C_dg a multiple of 32; spilling beyond 128 output channels) or,
// with 16-channel groups, left the 16-bit kernels altogether.
//
// W^T slabs beyond 48 KB (round 6: 256 input channels with >= 128 output channels, 128 with 256 -- 64 / 128 KB per
// tap) do not fit LDS next to the wave tiles of two workgroups.  Those instances (kWG) read the A fragments of the
// matrix phase STRAIGHT FROM GLOBAL MEMORY (the slab of all taps is L2-resident: 1.2 MB at 256 x 256 x 9) through a
// ring of kWRing fragments per wave that runs ahead of the MFMAs and across the tap boundary; no slab staging, and with it
// no workgroup barrier at all -- the four waves of a workgroup share nothing.  Before, such shapes ran on hp_bwd2, whose
// register-stationary W^T + grad_W accumulators spill at 16 k-steps (744 B of scratch per lane: the fp16 backward of
// MDCN2d 256 -> 256 at 56 x 56, B = 8 took 0.79 ms in that kernel alone, as long as the whole fp32 backward).
#pragma once
#include "hp_kernels.hpp"

namespace mdconv {

constexpr int kB3ChunkRows = 64;   // grad_out rows staged through LDS at a time (4 k-steps)
constexpr int kB3SlabMaxKB = 48;   // largest W^T slab of one tap that is staged in LDS
constexpr int kB3WRing2 = 8, kB3WRing3 = 4;   // kWG instances: A fragments in flight per wave (2-D, 3-D)

// LPP: lanes per pixel of a deformable group (C_dg / 8); CBT: 32-channel blocks of ALL channels (Cp / 32)
template <int ND, bool MOD, typename T, int LPP, int NKS, int CBT>
__global__ __launch_bounds__(256, 2) void hp_bwd3_kernel(
    Geom g, HpDims hd, const typename T::Raw *__restrict__ xt, const U4 *__restrict__ wpb,
    const typename T::Raw *__restrict__ gout, const typename T::Raw *__restrict__ offset,
    const typename T::Raw *__restrict__ mask, typename T::Raw *__restrict__ gcol,
    typename T::Raw *__restrict__ colbuf, typename T::Raw *__restrict__ grad_offset,
    typename T::Raw *__restrict__ grad_mask, int *__restrict__ cnt) {
  using Raw = typename T::Raw;
  constexpr int NC = 1 << ND;
  constexpr int SW = 2 * NC + 4;        // state dwords per (group, pixel): voff[NC] (later S[NC]), w*mask[NC], grad_col row, image, pad
  constexpr int Cp = CBT * 32;
  constexpr int NDG = Cp / (LPP * 8);   // deformable groups: 1, 2 or 4
  constexpr int NR = NDG > 1 ? NDG / 2 : 1;   // state-building rounds per tap (two groups per round)
  constexpr int PPI = 64 / LPP;         // pixels per gather iteration
  constexpr int NIT = 32 / PPI;         // gather iterations per (tap, group) unit
  constexpr int NSTEP = NDG * NIT;      // gather iterations per tap
  constexpr int CB = CBT;
  constexpr int pitch = Cp + 8;
  constexpr int WTOT = CB * NKS * 64;   // U4 elements of one tap's W^T slab
  constexpr int WPT = (WTOT + 255) / 256;        // ... per thread
  constexpr int WPI = (WPT + NSTEP - 1) / NSTEP; // ... per thread and gather iteration
  constexpr int REGION = (32 * pitch * 2 > kB3ChunkRows * kPP * 2 ? 32 * pitch * 2 : kB3ChunkRows * kPP * 2);
  constexpr bool kWG = CB * NKS > kB3SlabMaxKB;   // A fragments straight from global memory, no slab in LDS
  constexpr int NFR = CB * NKS;                   // A fragments (1 KB each) per tap
  constexpr int kWRing = ND == 2 ? kB3WRing2 : kB3WRing3;   // (3-D: 2^ND corner octets x two gather sets leave fewer registers)
  static_assert(NDG == 1 || NDG == 2 || NDG == 4, "1, 2 or 4 deformable groups");
  static_assert(NDG * LPP * 8 == Cp, "groups tile the channels");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  U4 *Ws = reinterpret_cast<U4 *>(smem);                                   // [CB][NKS][64] W^T fragments of the tap
  const int tid = threadIdx.x, lane = tid & 63, kh = lane >> 5, pl = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char *mine = smem + (kWG ? 0 : WTOT * 16) + wave * (REGION + NDG * 32 * SW * 4);
  Raw *Gc = reinterpret_cast<Raw *>(mine);                                 // [32][pitch]; first the grad_out staging tile
  int *St = reinterpret_cast<int *>(mine + REGION);                        // [NDG][32][SW]

  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  // first pixel of this wave's 32-pixel segment (wave-uniform; hp_common.hpp: linear or blocked tile order)
  int b0, p0;
  hp_wave_segment(g, hd.blocked, tile, wave, b0, p0);
  const bool wave_live = b0 < g.B;
  if (!wave_live) { b0 = 0; p0 = 0; }
  const bool one_img = p0 + 31 < g.S_o;             // all 32 pixels in image b0: scalar row bases, buffer stores

  // ---- the pixel this lane owns in the state role (lanes 32-63 mirror 0-31) ----
  int b = b0, pix = p0 + pl;
  while (pix >= g.S_o) { pix -= g.S_o; ++b; }
  const bool live = wave_live && b < g.B;
  if (!live) { b = g.B - 1; pix = g.S_o - 1; }
  int oc[ND];
  out_coords<ND>(g, pix, oc);

  const rsrc_t r_xt = make_rsrc(xt, (size_t)g.B * g.S_i * Cp * 2);
  const size_t gcol_img = (size_t)g.K * g.S_o * Cp;   // grad_col / column elements per image
  const rsrc_t r_gcol = make_rsrc(gcol + (size_t)b0 * gcol_img, gcol_img * 2);
  const rsrc_t r_col = make_rsrc(colbuf + (size_t)b0 * gcol_img, gcol_img * 2);
  const int S_e = hp_anchor_space(g);

  // ---- W^T slab of tap 0 -> LDS (whole workgroup) ----
  if (!kWG)
    for (int i = tid; i < WTOT; i += 256) Ws[i] = wpb[i];
  // kWG: ring of A fragments; fragment i of a tap lives in slot i % kWRing
  const rsrc_t r_w = make_rsrc(wpb, (size_t)g.K * WTOT * 16);
  U4 wring[kWG ? kWRing : 1];

  // ---- grad_out of the wave's 32 pixels -> B fragments (K = o, N = pixel), kB3ChunkRows rows at a time ----
  U4 gB[NKS];
  {
    const bool vec_ok = (g.S_o & 7) == 0;
    Raw *tl = Gc;   // [kB3ChunkRows][kPP]
#pragma unroll
    for (int c0 = 0; c0 < NKS * 16; c0 += kB3ChunkRows) {
      for (int item = lane; item < kB3ChunkRows * 4; item += 64) {
        const int o = c0 + (item >> 2), oct = item & 3;
        int bb = b0, pp = p0 + oct * 8;
        while (pp >= g.S_o) { pp -= g.S_o; ++bb; }
        U4 v = {0, 0, 0, 0};
        if (wave_live && o < g.O && bb < g.B) {
          if (vec_ok) {
            v = *reinterpret_cast<const U4 *>(gout + ((int64_t)bb * g.O + o) * g.S_o + pp);
          } else {
            unsigned short e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              e[j] = bb < g.B ? __builtin_bit_cast(unsigned short, gout[((int64_t)bb * g.O + o) * g.S_o + pp]) : (unsigned short)0;
              if (++pp == g.S_o) { pp = 0; ++bb; }
            }
            v.x = e[0] | ((u32)e[1] << 16); v.y = e[2] | ((u32)e[3] << 16);
            v.z = e[4] | ((u32)e[5] << 16); v.w = e[6] | ((u32)e[7] << 16);
          }
        }
        *reinterpret_cast<U4 *>(tl + (item >> 2) * kPP + oct * 8) = v;
      }
      // lane i of a 16-lane group addresses row (i >> 2), pixel quad (i & 3) of its 4 x 16 block
      const Raw *bp = tl + (8 * kh + ((lane & 15) >> 2)) * kPP + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
#pragma unroll
      for (int k = 0; k < kB3ChunkRows / 16; ++k)
        if (c0 / 16 + k < NKS) lds_tr2(bp + k * 16 * kPP, 4 * kPP, gB[c0 / 16 + k]);
    }
  }

  // ---- state role: offsets / mask one tap ahead ----
  // (raw 16-bit values until build(): a conversion inside fetch() would be a use of the load where it is issued, hp_fwd2.hip)
  // One group: lanes 0-31 / 32-63 handle taps t / t + 1 of their pixel (round r = 0 only).  Several groups: both
  // half-waves handle the SAME tap, groups 2r / 2r + 1 in round r.
  Raw dlr[NR][ND], mlr[NR];
  const Raw *off_px = offset + (int64_t)b * NDG * (ND * g.K) * g.S_o + pix;
  const Raw *msk_px = MOD ? mask + (int64_t)b * NDG * g.K * g.S_o + pix : nullptr;
  // the tap THIS lane builds next, its coordinates kept incrementally (one group: lanes 32-63 one tap ahead)
  int b_tap = NDG == 1 ? kh : 0, b_tcd[ND];
  {
    int t0[ND];
    tap_coords<ND>(g, min(b_tap, g.K - 1), t0);   // (lane-dependent only through kh: two integer divisions, once)
#pragma unroll
    for (int a = 0; a < ND; ++a) b_tcd[a] = t0[a];
  }
  auto advance = [&]() {   // to the next tap this lane builds: two taps on with one group, one tap on with several
    constexpr int steps = NDG == 1 ? 2 : 1;
    b_tap += steps;
#pragma unroll
    for (int i = 0; i < steps; ++i) {
      if (++b_tcd[ND - 1] == g.ksz[ND - 1]) {
        b_tcd[ND - 1] = 0;
        if (ND == 3) {
          if (++b_tcd[1] == g.ksz[1]) { b_tcd[1] = 0; ++b_tcd[0]; }
        } else {
          ++b_tcd[0];
        }
      }
    }
  };
  auto fetch = [&]() {   // offsets / mask of this lane's next states (clamped past the last tap: built, never used)
    const int tp = min(b_tap, g.K - 1);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int dg = NDG == 1 ? 0 : 2 * r + kh;
      const int64_t ch = (int64_t)dg * g.K + tp;
#pragma unroll
      for (int a = 0; a < ND; ++a) dlr[r][a] = off_px[(ch * ND + a) * g.S_o];
      if (MOD) mlr[r] = msk_px[ch * g.S_o];
    }
  };
  struct Fac { float wl[ND], wh[ND], sl[ND], sh[ND], mg; } fac[NR];
  int held[NDG == 1 ? SW : 1];   // one group: the state row lanes 32-63 built for the odd tap, until that tap starts
#pragma unroll
  for (int q = 0; q < (NDG == 1 ? SW : 1); ++q) held[q] = 0;
  auto store_row = [&](const int (&ev)[SW], int dg) {
    int *sp = St + (dg * 32 + pl) * SW;
#pragma unroll
    for (int q = 0; q < SW; q += 4) *reinterpret_cast<int4 *>(sp + q) = make_int4(ev[q], ev[q + 1], ev[q + 2], ev[q + 3]);
  };
  // sampling state of (tap, group dg, this lane's pixel) from dlr[r] / mlr[r] -> St row (or `held`); CSR counting
  auto build_state = [&](int r, int tap, int dg, const int *tcd, bool mine, bool hold) {
    float dl[ND], ml = 1.f;
#pragma unroll
    for (int a = 0; a < ND; ++a) dl[a] = T::ldf(&dlr[r][a]);
    if (MOD) ml = T::ldf(&mlr[r]);
    TapCoef<ND, float> tc;
    make_tap<ND, float>(g, oc, tcd, dl, true, tc);
    HpCorners<ND> hc;
    hp_corners<ND>(tc, hc);
    fac[r].mg = (!g.range_gate || tc.inside) ? ml : 0.f;
#pragma unroll
    for (int a = 0; a < ND; ++a) { fac[r].wl[a] = tc.wl[a]; fac[r].wh[a] = tc.wh[a]; fac[r].sl[a] = tc.sl[a]; fac[r].sh[a] = tc.sh[a]; }
    if (mine) {
      int ev[SW];
#pragma unroll
      for (int ci = 0; ci < NC; ++ci) {
        ev[ci] = (live && hc.idx[ci] >= 0) ? (b * g.S_i + hc.idx[ci]) * Cp * 2 : kHpOob;
        ev[NC + ci] = __float_as_int(live ? hc.w[ci] * ml : 0.f);
      }
      // grad_col / column row: byte offset inside its image's rows, and the image
      ev[2 * NC] = live ? (tap * g.S_o + pix) * Cp * 2 : kHpOob;
      ev[2 * NC + 1] = b;
      ev[2 * NC + 2] = ev[2 * NC + 3] = 0;
      if (NDG == 1 && hold) {
#pragma unroll
        for (int q = 0; q < (NDG == 1 ? SW : 1); ++q) held[q] = ev[q];
      } else {
        store_row(ev, dg);
      }
      if (live) {
        // scatter anchor of this sample (first pass of the CSR build, hp_col2im.hip)
        SampleAnchor<ND> sa;
        sample_anchor<ND>(g, tc, 1.f, sa);
        if (sa.on) atomicAdd(cnt + ((int64_t)b * NDG + dg) * S_e + sa.qa, 1);
      }
    }
  };
  // grad_offset / grad_mask of (tap, group dg, this lane's pixel) from the reduced S in its state row and fac[r]
  auto finish_one = [&](int r, int tap, int dg) {
    float S[NC];
    const int *sp = St + (dg * 32 + pl) * SW;
#pragma unroll
    for (int q = 0; q < NC; q += 4) {
      const int4 e = *reinterpret_cast<const int4 *>(sp + q);
      S[q] = __int_as_float(e.x); S[q + 1] = __int_as_float(e.y); S[q + 2] = __int_as_float(e.z); S[q + 3] = __int_as_float(e.w);
    }
    float gm = 0.f, goff[ND];
#pragma unroll
    for (int a = 0; a < ND; ++a) goff[a] = 0.f;
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) {
      float w = 1.f;
#pragma unroll
      for (int a = 0; a < ND; ++a) w *= ((ci >> (ND - 1 - a)) & 1) ? fac[r].wh[a] : fac[r].wl[a];
      gm = fmaf(w, S[ci], gm);
#pragma unroll
      for (int a = 0; a < ND; ++a) {
        float dw = 1.f;
#pragma unroll
        for (int a2 = 0; a2 < ND; ++a2) {
          const bool hi = (ci >> (ND - 1 - a2)) & 1;
          dw *= (a2 == a) ? (hi ? fac[r].sh[a2] : fac[r].sl[a2]) : (hi ? fac[r].wh[a2] : fac[r].wl[a2]);
        }
        goff[a] = fmaf(dw, S[ci], goff[a]);
      }
    }
    const int64_t ch = (int64_t)dg * g.K + tap;
    Raw *go = grad_offset + (int64_t)b * NDG * (ND * g.K) * g.S_o + pix + ch * ND * g.S_o;
#pragma unroll
    for (int a = 0; a < ND; ++a) {
      Raw *d = go + (int64_t)a * g.S_o;
      T::stf(d, goff[a] * fac[r].mg + (g.acc_data ? T::ldf(d) : 0.f));
    }
    if (MOD) {
      Raw *d = grad_mask + (int64_t)b * NDG * g.K * g.S_o + pix + ch * g.S_o;
      T::stf(d, gm + (g.acc_data ? T::ldf(d) : 0.f));
    }
  };
  auto finish = [&](int tap) {
    if (NDG == 1) {
      if (kh == (tap & 1) && live) finish_one(0, tap, 0);
    } else if (live) {
#pragma unroll
      for (int r = 0; r < NR; ++r) finish_one(r, tap, 2 * r + kh);
    }
  };

  // ---- gather role: step j of a tap = iteration j % NIT of the group j / NIT ----
  const int gp = lane / LPP, oc8 = lane % LPP;
  // a Set holds only the gathered corner octets; weights, row offset and the grad_col piece are re-read
  // from LDS when the set is consumed (28 registers less per wave: two sets in flight fit without spills)
  struct Set { U4 x[NC]; };
  auto issue = [&](Set &s, int j) {
    const int dg = j / NIT, it = j - dg * NIT;
    const int *sp = St + (dg * 32 + it * PPI + gp) * SW;
    int ev[NC];
#pragma unroll
    for (int q = 0; q < NC; q += 4) {
      const int4 e = *reinterpret_cast<const int4 *>(sp + q);
      ev[q] = e.x; ev[q + 1] = e.y; ev[q + 2] = e.z; ev[q + 3] = e.w;
    }
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) {
      s.x[ci] = buf_load4u(r_xt, ev[ci] + (dg * LPP + oc8) * 16, 0);
    }
  };
  auto consume = [&](const Set &s, int j) {
    const int dg = j / NIT, it = j - dg * NIT;
    const int p = it * PPI + gp;
    const int cho = (dg * LPP + oc8) * 8;   // this lane's channel octet
    int *sp = St + (dg * 32 + p) * SW;
    float w[NC];
#pragma unroll
    for (int q = 0; q < NC; q += 4) {
      const int4 e = *reinterpret_cast<const int4 *>(sp + NC + q);
      w[q] = __int_as_float(e.x); w[q + 1] = __int_as_float(e.y); w[q + 2] = __int_as_float(e.z); w[q + 3] = __int_as_float(e.w);
    }
    const int4 tail = *reinterpret_cast<const int4 *>(sp + 2 * NC);   // (grad_col row offset, image, -, -)
    const U4 gq = *reinterpret_cast<const U4 *>(Gc + p * pitch + cho);
    float S[NC];
    U4 cq = {0, 0, 0, 0};
    if constexpr (T::kPackedCol) {
#pragma unroll
      for (int ci = 0; ci < NC; ++ci) {
        S[ci] = dot8<T>(0.f, s.x[ci], gq);
        T::pk_mac8(cq, s.x[ci], w[ci]);
      }
    } else {
      float col[8];
#pragma unroll
      for (int j2 = 0; j2 < 8; ++j2) col[j2] = 0.f;
#pragma unroll
      for (int ci = 0; ci < NC; ++ci) {
        S[ci] = dot8<T>(0.f, s.x[ci], gq);
        mac8<T>(col, s.x[ci], w[ci]);
      }
      cq = pack8<T>(col);
    }
    if (one_img) {   // scalar row base, dead pixels out of range (dropped)
      buf_store4u_nt(r_gcol, tail.x + cho * 2, 0, gq);
      buf_store4u_nt(r_col, tail.x + cho * 2, 0, cq);
    } else if (tail.x != kHpOob) {
      const size_t e = (size_t)tail.y * gcol_img + (tail.x >> 1) + cho;
      *reinterpret_cast<U4 *>(gcol + e) = gq;
      *reinterpret_cast<U4 *>(colbuf + e) = cq;
    }
    // S summed over the pixel's LPP lanes: DPP adds inside a row of 16, ds_bpermute beyond
    hp_dpp_sum<NC>(S, LPP < 16 ? LPP : 16);
    if (LPP == 32) {
#pragma unroll
      for (int ci = 0; ci < NC; ++ci) S[ci] += __shfl_xor(S[ci], 16, 64);
    }
    if (oc8 == 0) {   // the row's corner offsets are dead by now: S takes their place
#pragma unroll
      for (int q = 0; q < NC; q += 4)
        *reinterpret_cast<int4 *>(sp + q) = make_int4(__float_as_int(S[q]), __float_as_int(S[q + 1]),
                                                      __float_as_int(S[q + 2]), __float_as_int(S[q + 3]));
    }
  };

  fetch();
  if (kWG) {
    static_assert(!kWG || NFR % kWRing == 0, "fragments per tap: a whole number of ring turns");
#pragma unroll
    for (int i = 0; i < (kWG ? kWRing : 0); ++i) wring[i] = buf_load4u(r_w, lane * 16, i * 1024);
  } else {
    __syncthreads();   // W^T slab of tap 0 is in LDS (and every wave is past its grad_out staging)
  }
  for (int tap = 0; tap < g.K; ++tap) {
    if (wave_live) {
      if (NDG == 1) {
        if ((tap & 1) == 0) {
          build_state(0, b_tap, 0, b_tcd, b_tap < g.K, kh != 0);
          advance();
          if (tap + 2 < g.K) fetch();
        } else if (kh) {
          int ev[SW];
#pragma unroll
          for (int q = 0; q < SW; ++q) ev[q] = held[NDG == 1 ? q : 0];
          store_row(ev, 0);
        }
      } else {
#pragma unroll
        for (int r = 0; r < NR; ++r) build_state(r, tap, 2 * r + kh, b_tcd, true, false);
        advance();
        if (tap + 1 < g.K) fetch();
      }
      // ---- matrix phase: GEMM-1 per 32-channel block -> Gc ----
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if constexpr (kWG) {
          // fragment i is consumed from its ring slot, which is refilled with fragment i + kWRing -- of the NEXT tap
          // past the end of this one (clamped on the last tap: loaded, never used), so the first fragments of a tap
          // arrive during the gather phase before it
          const int t_nxt = tap + 1 < g.K ? tap + 1 : tap;
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) {
            const int i = cb * NKS + ks;
            acc = T::mfma(wring[i % kWRing], gB[ks], acc);
            wring[i % kWRing] = i + kWRing < NFR ? buf_load4u(r_w, lane * 16, (tap * NFR + i + kWRing) * 1024)
                                                 : buf_load4u(r_w, lane * 16, (t_nxt * NFR + i + kWRing - NFR) * 1024);
          }
        } else {
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) acc = T::mfma(Ws[(cb * NKS + ks) * 64 + lane], gB[ks], acc);
        }
        float g0[8], g1[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { g0[r] = acc[r]; g1[r] = acc[8 + r]; }
        Raw *dst = Gc + pl * pitch + cb * 32 + 16 * kh;
        *reinterpret_cast<U4 *>(dst) = pack8<T>(g0);
        *reinterpret_cast<U4 *>(dst + 8) = pack8<T>(g1);
      }
    }
    if (!kWG) __syncthreads();   // B1: every wave is done with this tap's W^T slab
    // ---- gather phase (+ the next tap's W^T slab, WPI pieces per iteration) ----
    // piece k of iteration `j`: element tid + (j * WPI + k) * 256 of the slab, loaded at the top of the
    // iteration (unconditionally, from a clamped index: a conditional load into a struct ended up in scratch)
    // and stored to LDS at its end
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const bool stage = !kWG && tap + 1 < g.K;
    const u32x4 *wsrc = reinterpret_cast<const u32x4 *>(wpb + (size_t)(stage ? tap + 1 : tap) * WTOT);
    u32x4 *wdst = reinterpret_cast<u32x4 *>(Ws);
    Set sa, sb;
    if (wave_live) issue(sa, 0);
#pragma unroll
    for (int j = 0; j < NSTEP; ++j) {
      static_assert(WPI <= 2, "W^T pieces per gather iteration");
      const int i0 = tid + (j * WPI) * 256, i1 = i0 + 256;
      const bool on0 = stage && j * WPI < WPT && i0 < WTOT;
      const bool on1 = WPI > 1 && stage && j * WPI + 1 < WPT && i1 < WTOT;
      const u32x4 w0 = wsrc[on0 ? i0 : 0];
      const u32x4 w1 = wsrc[on1 ? i1 : 0];
      if (wave_live) {
        if (j & 1) {
          if (j + 1 < NSTEP) issue(sa, j + 1);
          consume(sb, j);
        } else {
          if (j + 1 < NSTEP) issue(sb, j + 1);
          consume(sa, j);
        }
      }
      if (on0) wdst[i0] = w0;
      if (on1) wdst[i1] = w1;
    }
    if (wave_live) finish(tap);
    if (!kWG) __syncthreads();   // B2: the next tap's W^T slab is complete
  }
}

size_t hp_bwd3_lds_bytes(const Geom &g, const HpDims &hd);   // hp_bwd3.hip

template <int ND, bool MOD, typename T, int LPP, int NKS, int CBT>
int launch_bwd3(const Geom &g, const HpDims &hd, const Tensors &t, const void *xt, const void *wpb,
                void *gcol, void *colbuf, int *cnt, hipStream_t stream) {
  using Raw = typename T::Raw;
  const size_t lds = hp_bwd3_lds_bytes(g, hd);
  if (lds > 64 * 1024) {
    hipError_t ea = hipFuncSetAttribute((const void *)hp_bwd3_kernel<ND, MOD, T, LPP, NKS, CBT>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ea != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(ea)); return MDCONV_ELAUNCH; }
  }
  hp_debug_plan("hp_bwd3", hp_bwd3_kernel<ND, MOD, T, LPP, NKS, CBT>, 256, lds, (g.N + 127) / 128);
  hipLaunchKernelGGL((hp_bwd3_kernel<ND, MOD, T, LPP, NKS, CBT>), dim3((g.N + 127) / 128), dim3(256), lds, stream, g, hd,
                     (const Raw *)xt, (const U4 *)wpb, (const Raw *)t.grad_output, (const Raw *)t.offset,
                     (const Raw *)t.mask, (Raw *)gcol, (Raw *)colbuf, (Raw *)t.grad_offset, (Raw *)t.grad_mask, cnt);
  return check_launch("hp_bwd3");
}

// (Cp, deformable groups) -> (LPP, CBT): LPP = C_dg / 8 lanes per pixel of a group, CBT = Cp / 32
template <int ND, bool MOD, typename T>
int dispatch_bwd3(const Geom &g, const HpDims &hd, const Tensors &t, const void *xt, const void *wpb,
                  void *gcol, void *colbuf, int *cnt, hipStream_t stream) {
#define HP_B3(L, N, C) return launch_bwd3<ND, MOD, T, L, N, C>(g, hd, t, xt, wpb, gcol, colbuf, cnt, stream)
#define HP_B3_L(L, C)                                                              \
  switch (hd.nks) {                                                                \
    case 2: HP_B3(L, 2, C);                                                        \
    case 4: HP_B3(L, 4, C);                                                        \
    case 8: HP_B3(L, 8, C);                                                        \
    default: HP_B3(L, 16, C);                                                      \
  }
  const int lpp = hd.Cp / g.DG / 8;
  switch (hd.Cp) {
    case 32:
      if (lpp == 4) HP_B3_L(4, 1);
      HP_B3_L(2, 1);
    case 64:
      if (lpp == 8) HP_B3_L(8, 2);
      if (lpp == 4) HP_B3_L(4, 2);
      HP_B3_L(2, 2);
    case 128:
      if (lpp == 16) HP_B3_L(16, 4);
      if (lpp == 8) HP_B3_L(8, 4);
      HP_B3_L(4, 4);
    default:
      if (lpp == 32) HP_B3_L(32, 8);
      if (lpp == 16) HP_B3_L(16, 8);
      HP_B3_L(8, 8);
  }
#undef HP_B3_L
#undef HP_B3
  return MDCONV_EUNSUPPORTED;   // (not reached: hp_bwd3_supported admits only the combinations above)
}

}  // namespace mdconv
