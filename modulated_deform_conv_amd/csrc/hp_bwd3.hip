// hp_bwd3.hip -- pixel-stationary backward kernel of the native 16-bit path (gfx950).
//
// Reference: mdeformable_conv.cu:412-444, 202-318; 3-D mdeformable_conv3d.cu:515-560, 265-395.
// Per (tap, pixel)
//     GEMM-1  gc[c] = sum_o W[o, c, tap] grad_out[o, n]
//     S[ci]   = sum_c gc[c] x[ci][c]  -> grad_mask, grad_offset;   grad_col row -> workspace
//     col[c]  = mask * sum_ci w[ci] x[ci][c]                       column row   -> workspace
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
//   * gather phase, lane = (pixel, channel octet), Cp/8 ADJACENT lanes per pixel (line-wide gathers,
//     tools/ubench_gather16.hip): the grad_col piece goes to the workspace, the 2^ND corner octets are
//     gathered (two iterations in flight), S accumulates with v_dot2c, col with v_fma_mix, S is reduced
//     over the pixel's lanes with DPP adds and parked in the pixel's state row;
//   * lanes 0-31 = the wave's pixels build the sampling state of every tap (offsets / mask one tap
//     ahead, CSR counting atomic) and finish grad_offset / grad_mask from S (single owner, no atomics).
// Shapes: one conv group, one deformable group, Cp in {32, 64, 128, 256}, W^T slab <= 48 KB; everything
// else stays on hp_bwd2 / hp_bwd.
#include "hp_kernels.hpp"

namespace mdconv {

namespace {


constexpr int kChunkRows = 64;   // grad_out rows staged through LDS at a time (4 k-steps)
// Sampling states are built TWO taps at a time (round 5): at an even tap t lanes 0-31 build (t, pixel) and lanes
// 32-63 build (t + 1, pixel) -- before, both half-waves computed the same state and one discarded it.  The state
// table has one row set per wave, so lanes 32-63 keep their row in registers (18 dwords) until tap t + 1 starts;
// the per-axis factors `fac` stay in the half-wave that built them, which is also the one that finishes the tap's
// grad_offset / grad_mask.

template <int ND, bool MOD, typename T, int LPP, int NKS>
__global__ __launch_bounds__(256, 2) void hp_bwd3_kernel(
    Geom g, HpDims hd, const typename T::Raw *__restrict__ xt, const U4 *__restrict__ wpb,
    const typename T::Raw *__restrict__ gout, const typename T::Raw *__restrict__ offset,
    const typename T::Raw *__restrict__ mask, typename T::Raw *__restrict__ gcol,
    typename T::Raw *__restrict__ colbuf, typename T::Raw *__restrict__ grad_offset,
    typename T::Raw *__restrict__ grad_mask, int *__restrict__ cnt) {
  using Raw = typename T::Raw;
  constexpr int NC = 1 << ND;
  constexpr int SW = 2 * NC + 4;        // state dwords per pixel: voff[NC] (later S[NC]), w*mask[NC], grad_col row, pad
  constexpr int PPI = 64 / LPP;         // pixels per gather iteration
  constexpr int NIT = 32 / PPI;         // gather iterations per tap
  constexpr int CB = LPP / 4;           // 32-channel blocks
  constexpr int Cp = LPP * 8;
  constexpr int pitch = Cp + 8;
  constexpr int WTOT = CB * NKS * 64;   // U4 elements of one tap's W^T slab
  constexpr int WPT = (WTOT + 255) / 256;        // ... per thread
  constexpr int WPI = (WPT + NIT - 1) / NIT;     // ... per thread and gather iteration
  constexpr int REGION = (32 * pitch * 2 > kChunkRows * kPP * 2 ? 32 * pitch * 2 : kChunkRows * kPP * 2);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  U4 *Ws = reinterpret_cast<U4 *>(smem);                                   // [CB][NKS][64] W^T fragments of the tap
  const int tid = threadIdx.x, lane = tid & 63, kh = lane >> 5, pl = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char *mine = smem + WTOT * 16 + wave * (REGION + 32 * SW * 4);
  Raw *Gc = reinterpret_cast<Raw *>(mine);                                 // [32][pitch]; first the grad_out staging tile
  int *St = reinterpret_cast<int *>(mine + REGION);                        // [32][SW]

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
  for (int i = tid; i < WTOT; i += 256) Ws[i] = wpb[i];

  // ---- grad_out of the wave's 32 pixels -> B fragments (K = o, N = pixel), kChunkRows rows at a time ----
  U4 gB[NKS];
  {
    const bool vec_ok = (g.S_o & 7) == 0;
    Raw *tl = Gc;   // [kChunkRows][kPP]
#pragma unroll
    for (int c0 = 0; c0 < NKS * 16; c0 += kChunkRows) {
      for (int item = lane; item < kChunkRows * 4; item += 64) {
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
      for (int k = 0; k < kChunkRows / 16; ++k)
        if (c0 / 16 + k < NKS) lds_tr2(bp + k * 16 * kPP, 4 * kPP, gB[c0 / 16 + k]);
    }
  }

  // ---- state role: offsets / mask one tap ahead ----
  // (raw 16-bit values until build(): a conversion inside fetch() would be a use of the load where it is issued, hp_fwd2.hip)
  Raw dlr[ND], mlr;
  const Raw *off_px = offset + (int64_t)b * (ND * g.K) * g.S_o + pix;
  const Raw *msk_px = MOD ? mask + (int64_t)b * g.K * g.S_o + pix : nullptr;
  // the tap THIS lane builds next (lanes 32-63 one ahead), its coordinates kept incrementally
  int b_tap = kh, b_tcd[ND];
  {
    int t0[ND];
    tap_coords<ND>(g, min(kh, g.K - 1), t0);   // (lane-dependent only through kh: two integer divisions, once)
#pragma unroll
    for (int a = 0; a < ND; ++a) b_tcd[a] = t0[a];
  }
  auto advance2 = [&]() {
    b_tap += 2;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
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
  auto fetch = [&]() {   // offsets / mask of this lane's next tap (clamped past the last: built, never used)
    const int tp = min(b_tap, g.K - 1);
#pragma unroll
    for (int a = 0; a < ND; ++a) dlr[a] = off_px[((int64_t)tp * ND + a) * g.S_o];
    if (MOD) mlr = msk_px[(int64_t)tp * g.S_o];
  };
  struct Fac { float wl[ND], wh[ND], sl[ND], sh[ND], mg; } fac;
  int held[SW];   // the state row lanes 32-63 built for the odd tap, until that tap starts
#pragma unroll
  for (int q = 0; q < SW; ++q) held[q] = 0;
  auto store_row = [&](const int (&ev)[SW]) {
    int *sp = St + pl * SW;
#pragma unroll
    for (int q = 0; q < SW; q += 4) *reinterpret_cast<int4 *>(sp + q) = make_int4(ev[q], ev[q + 1], ev[q + 2], ev[q + 3]);
  };
  // sampling state of (tap, this lane's pixel) from dl / ml -> St row (or `held`); CSR counting
  auto build_state = [&](int tap, const int *tcd, bool mine, bool hold) {
    float dl[ND], ml = 1.f;
#pragma unroll
    for (int a = 0; a < ND; ++a) dl[a] = T::ldf(&dlr[a]);
    if (MOD) ml = T::ldf(&mlr);
    TapCoef<ND, float> tc;
    make_tap<ND, float>(g, oc, tcd, dl, true, tc);
    HpCorners<ND> hc;
    hp_corners<ND>(tc, hc);
    fac.mg = (!g.range_gate || tc.inside) ? ml : 0.f;
#pragma unroll
    for (int a = 0; a < ND; ++a) { fac.wl[a] = tc.wl[a]; fac.wh[a] = tc.wh[a]; fac.sl[a] = tc.sl[a]; fac.sh[a] = tc.sh[a]; }
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
      if (hold) {
#pragma unroll
        for (int q = 0; q < SW; ++q) held[q] = ev[q];
      } else {
        store_row(ev);
      }
      if (live) {
        // scatter anchor of this sample (first pass of the CSR build, hp_col2im.hip)
        SampleAnchor<ND> sa;
        sample_anchor<ND>(g, tc, 1.f, sa);
        if (sa.on) atomicAdd(cnt + (int64_t)b * S_e + sa.qa, 1);
      }
    }
  };
  auto finish = [&](int tap) {   // grad_offset / grad_mask of (tap, this lane's pixel) from the reduced S in its state row
    if (kh == (tap & 1) && live) {
      float S[NC];
      const int *sp = St + pl * SW;
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
        for (int a = 0; a < ND; ++a) w *= ((ci >> (ND - 1 - a)) & 1) ? fac.wh[a] : fac.wl[a];
        gm = fmaf(w, S[ci], gm);
#pragma unroll
        for (int a = 0; a < ND; ++a) {
          float dw = 1.f;
#pragma unroll
          for (int a2 = 0; a2 < ND; ++a2) {
            const bool hi = (ci >> (ND - 1 - a2)) & 1;
            dw *= (a2 == a) ? (hi ? fac.sh[a2] : fac.sl[a2]) : (hi ? fac.wh[a2] : fac.wl[a2]);
          }
          goff[a] = fmaf(dw, S[ci], goff[a]);
        }
      }
      Raw *go = grad_offset + (int64_t)b * (ND * g.K) * g.S_o + pix + (int64_t)tap * ND * g.S_o;
#pragma unroll
      for (int a = 0; a < ND; ++a) {
        Raw *d = go + (int64_t)a * g.S_o;
        T::stf(d, goff[a] * fac.mg + (g.acc_data ? T::ldf(d) : 0.f));
      }
      if (MOD) {
        Raw *d = grad_mask + (int64_t)b * g.K * g.S_o + pix + (int64_t)tap * g.S_o;
        T::stf(d, gm + (g.acc_data ? T::ldf(d) : 0.f));
      }
    }
  };

  // ---- gather role ----
  const int gp = lane / LPP, oc8 = lane % LPP;
  // a Set holds only the gathered corner octets; weights, row offset and the grad_col piece are re-read
  // from LDS when the set is consumed (28 registers less per wave: two sets in flight fit without spills)
  struct Set { U4 x[NC]; };
  auto issue = [&](Set &s, int it) {
    const int *sp = St + (it * PPI + gp) * SW;
    int ev[NC];
#pragma unroll
    for (int q = 0; q < NC; q += 4) {
      const int4 e = *reinterpret_cast<const int4 *>(sp + q);
      ev[q] = e.x; ev[q + 1] = e.y; ev[q + 2] = e.z; ev[q + 3] = e.w;
    }
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) {
      s.x[ci] = buf_load4u(r_xt, ev[ci] + oc8 * 16, 0);
    }
  };
  auto consume = [&](const Set &s, int it) {
    const int p = it * PPI + gp;
    int *sp = St + p * SW;
    float w[NC];
#pragma unroll
    for (int q = 0; q < NC; q += 4) {
      const int4 e = *reinterpret_cast<const int4 *>(sp + NC + q);
      w[q] = __int_as_float(e.x); w[q + 1] = __int_as_float(e.y); w[q + 2] = __int_as_float(e.z); w[q + 3] = __int_as_float(e.w);
    }
    const int4 tail = *reinterpret_cast<const int4 *>(sp + 2 * NC);   // (grad_col row offset, image, -, -)
    const U4 gq = *reinterpret_cast<const U4 *>(Gc + p * pitch + oc8 * 8);
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
      for (int j = 0; j < 8; ++j) col[j] = 0.f;
#pragma unroll
      for (int ci = 0; ci < NC; ++ci) {
        S[ci] = dot8<T>(0.f, s.x[ci], gq);
        mac8<T>(col, s.x[ci], w[ci]);
      }
      cq = pack8<T>(col);
    }
    if (one_img) {   // scalar row base, dead pixels out of range (dropped)
      buf_store4u_nt(r_gcol, tail.x + oc8 * 16, 0, gq);
      buf_store4u_nt(r_col, tail.x + oc8 * 16, 0, cq);
    } else if (tail.x != kHpOob) {
      const size_t e = (size_t)tail.y * gcol_img + (tail.x >> 1) + oc8 * 8;
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
  __syncthreads();   // W^T slab of tap 0 is in LDS (and every wave is past its grad_out staging)
  for (int tap = 0; tap < g.K; ++tap) {
    if (wave_live) {
      if ((tap & 1) == 0) {
        build_state(b_tap, b_tcd, b_tap < g.K, kh != 0);
        advance2();
        if (tap + 2 < g.K) fetch();
      } else if (kh) {
        store_row(held);
      }
      // ---- matrix phase: GEMM-1 per 32-channel block -> Gc ----
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) acc = T::mfma(Ws[(cb * NKS + ks) * 64 + lane], gB[ks], acc);
        float g0[8], g1[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { g0[r] = acc[r]; g1[r] = acc[8 + r]; }
        Raw *dst = Gc + pl * pitch + cb * 32 + 16 * kh;
        *reinterpret_cast<U4 *>(dst) = pack8<T>(g0);
        *reinterpret_cast<U4 *>(dst + 8) = pack8<T>(g1);
      }
    }
    __syncthreads();   // B1: every wave is done with this tap's W^T slab
    // ---- gather phase (+ the next tap's W^T slab, WPI pieces per iteration) ----
    // piece k of iteration `it`: element tid + (it * WPI + k) * 256 of the slab, loaded at the top of the
    // iteration (unconditionally, from a clamped index: a conditional load into a struct ended up in scratch)
    // and stored to LDS at its end
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const bool stage = tap + 1 < g.K;
    const u32x4 *wsrc = reinterpret_cast<const u32x4 *>(wpb + (size_t)(stage ? tap + 1 : tap) * WTOT);
    u32x4 *wdst = reinterpret_cast<u32x4 *>(Ws);
    Set sa, sb;
    if (wave_live) issue(sa, 0);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      static_assert(WPI <= 2, "W^T pieces per gather iteration");
      const int i0 = tid + (it * WPI) * 256, i1 = i0 + 256;
      const bool on0 = stage && it * WPI < WPT && i0 < WTOT;
      const bool on1 = WPI > 1 && stage && it * WPI + 1 < WPT && i1 < WTOT;
      const u32x4 w0 = wsrc[on0 ? i0 : 0];
      const u32x4 w1 = wsrc[on1 ? i1 : 0];
      if (wave_live) {
        if (it & 1) {
          if (it + 1 < NIT) issue(sa, it + 1);
          consume(sb, it);
        } else {
          if (it + 1 < NIT) issue(sb, it + 1);
          consume(sa, it);
        }
      }
      if (on0) wdst[i0] = w0;
      if (on1) wdst[i1] = w1;
    }
    if (wave_live) finish(tap);
    __syncthreads();   // B2: the next tap's W^T slab is complete
  }
}

}  // namespace


size_t hp_bwd3_lds_bytes(const HpDims &hd) {
  const size_t region = (size_t)32 * (hd.Cp + 8) * 2 > (size_t)kChunkRows * kPP * 2 ? (size_t)32 * (hd.Cp + 8) * 2
                                                                                    : (size_t)kChunkRows * kPP * 2;
  return (size_t)hd.cblks * hd.nks * 1024 + 4 * (region + 32 * (2 * 8 + 4) * 4);
}

bool hp_bwd3_supported(const Geom &g, const HpDims &hd) {
  if (g.G != 1 || g.DG != 1) return false;
  if (hd.Cp != 32 && hd.Cp != 64 && hd.Cp != 128 && hd.Cp != 256) return false;
  if (hd.nks > 16 || hd.cblks * hd.nks > 48) return false;   // W^T slab of one tap <= 48 KB
  return hp_bwd3_lds_bytes(hd) <= 80 * 1024;                  // two workgroups per CU
}

template <int ND, bool MOD, typename T, int LPP, int NKS>
static int launch_bwd3(const Geom &g, const HpDims &hd, const Tensors &t, const void *xt, const void *wpb,
                       void *gcol, void *colbuf, int *cnt, hipStream_t stream) {
  using Raw = typename T::Raw;
  const size_t lds = hp_bwd3_lds_bytes(hd);
  if (lds > 64 * 1024) {
    hipError_t ea = hipFuncSetAttribute((const void *)hp_bwd3_kernel<ND, MOD, T, LPP, NKS>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ea != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(ea)); return MDCONV_ELAUNCH; }
  }
  hp_debug_plan("hp_bwd3", hp_bwd3_kernel<ND, MOD, T, LPP, NKS>, 256, lds, (g.N + 127) / 128);
  hipLaunchKernelGGL((hp_bwd3_kernel<ND, MOD, T, LPP, NKS>), dim3((g.N + 127) / 128), dim3(256), lds, stream, g, hd,
                     (const Raw *)xt, (const U4 *)wpb, (const Raw *)t.grad_output, (const Raw *)t.offset,
                     (const Raw *)t.mask, (Raw *)gcol, (Raw *)colbuf, (Raw *)t.grad_offset, (Raw *)t.grad_mask, cnt);
  return check_launch("hp_bwd3");
}

template <int ND, bool MOD, typename T>
static int dispatch_bwd3(const Geom &g, const HpDims &hd, const Tensors &t, const void *xt, const void *wpb,
                         void *gcol, void *colbuf, int *cnt, hipStream_t stream) {
#define HP_B3(L, N) return launch_bwd3<ND, MOD, T, L, N>(g, hd, t, xt, wpb, gcol, colbuf, cnt, stream)
#define HP_B3_L(L)                                                                 \
  switch (hd.nks) {                                                                \
    case 2: HP_B3(L, 2);                                                           \
    case 4: HP_B3(L, 4);                                                           \
    case 8: HP_B3(L, 8);                                                           \
    default: HP_B3(L, 16);                                                         \
  }
  switch (hd.Cp) {
    case 32: HP_B3_L(4);
    case 64: HP_B3_L(8);
    case 128: HP_B3_L(16);
    default: HP_B3_L(32);
  }
#undef HP_B3_L
#undef HP_B3
}

int hp_backward3_launch(const Geom &g, const HpDims &hd, int dtype, const Tensors &t, const void *xt,
                        const void *wpb, void *gcol, void *colbuf, int *cnt, hipStream_t stream) {
#define HP_DISPATCH(T)                                                                          \
  do {                                                                                          \
    if (g.nd == 2)                                                                              \
      return g.modulated ? dispatch_bwd3<2, true, T>(g, hd, t, xt, wpb, gcol, colbuf, cnt, stream)    \
                         : dispatch_bwd3<2, false, T>(g, hd, t, xt, wpb, gcol, colbuf, cnt, stream);  \
    return g.modulated ? dispatch_bwd3<3, true, T>(g, hd, t, xt, wpb, gcol, colbuf, cnt, stream)      \
                       : dispatch_bwd3<3, false, T>(g, hd, t, xt, wpb, gcol, colbuf, cnt, stream);    \
  } while (0)
  if (dtype == MDCONV_F16) HP_DISPATCH(F16);
  HP_DISPATCH(BF16);
#undef HP_DISPATCH
}

}  // namespace mdconv
