// hp_fwd2.hip -- forward for fp16 / bf16 tensors, gather layout tuned to the texture path (gfx950).
//
// Same contraction as hp_fwd.hip (reference: mdeformable_conv.cu:37-87, 172-183), different
// gather.  Measured on MI355X (tools/ubench_gather16.hip): a 64-lane 16-byte load costs ~17 cycles
// when every aligned quad of lanes reads 64 contiguous bytes and ~66 cycles when the 16-byte
// pieces of a line sit 16 or 32 lanes apart -- which is what the MFMA-native mapping of
// hp_fwd.hip (lane = pixel, half-wave = channel octet) does.  So here a pixel's corner is read by
// 8 ADJACENT lanes (8 x 16 B = one 128-byte line = 64 channels), a wave-load covers 8 pixels, and
// the interpolated values take a short trip through a wave-private LDS tile to reach the B-fragment
// layout (one ds_write_b128 + one ds_read_b128 per 8 channels of a pixel -- no barrier, the wave
// reads what it wrote).  Everything else is as in hp_fwd.hip: a wave owns 32 pixels and all output
// channels of the workgroup, weights are staged global -> LDS once per workgroup and K stage
// (64 channels of one tap), fp32 interpolation with v_fma_mix_f32.
// The sampling state of (tap, pixel) -- 2^ND corner byte offsets (or the out-of-range marker) and
// 2^ND weights with validity and mask folded in -- is computed once by lanes 0-31 and parked in
// LDS; the 8 lanes of a pixel read it back as broadcasts.
#include "hp_kernels.hpp"

namespace mdconv {

namespace {


constexpr int kStage = 4;    // 16-channel chunks per K stage (= 64 channels = one 128-byte line)
// Sampling states are built TWO at a time (round 5): lanes 0-31 build state s of their pixel, lanes 32-63 state s + 1
// (the next tap, or the next deformable group of the tap), instead of both half-waves computing the same state and
// one of them discarding it.  The table has three slots per wave (state s lives in slot s % 3): when the pair
// (s + 2, s + 3) is built, during the last stage of state s + 1, the slots of s + 2 and of s (done) are free.
constexpr int kStSlots = 3;
constexpr int kBtP = 72;     // LDS pitch (16-bit elements) of a pixel row of the B tile: 64 + 8

// GRP = false: one conv group -- every chunk feeds every output-channel block, no table lookups.
template <int ND, bool MOD, typename T, int MB, bool GRP>
__global__ __launch_bounds__(256, 2) void hp_fwd2_kernel(
    Geom g, HpDims hd, const typename T::Raw *__restrict__ xt, const U4 *__restrict__ wpf,
    const typename T::Raw *__restrict__ bias, const typename T::Raw *__restrict__ offset,
    const typename T::Raw *__restrict__ mask, typename T::Raw *__restrict__ output,
    const int2 *__restrict__ ctab) {
  using Raw = typename T::Raw;
  constexpr int NC = 1 << ND;
  constexpr int SW = 2 * NC;   // state dwords per pixel
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nmax = GRP ? hd.fwd_nmax : MB;
  U4 *As = reinterpret_cast<U4 *>(smem);                                   // [2][kStage][nmax][64]
  Raw *Bt_all = reinterpret_cast<Raw *>(As + 2 * kStage * nmax * 64);      // [4][32][kBtP]
  int *St_all = reinterpret_cast<int *>(Bt_all + 4 * 32 * kBtP);           // [4][kStSlots][32][SW]

  const int tid = threadIdx.x, lane = tid & 63, kh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gp = lane >> 3, oc8 = lane & 7;   // gather role: pixel within a group of 8, channel octet
  Raw *Bt = Bt_all + wave * 32 * kBtP;
  int *St = St_all + wave * kStSlots * 32 * SW;
  const int orange = blockIdx.y;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int nchunks = hd.Cp / 16;
  const int2 *ct = ctab + orange * (nchunks + 1);
  int ch_lo = 0, ch_hi = (g.C + 15) / 16;
  if (GRP) {
    const int2 rng = ct[nchunks];
    ch_lo = rng.x & ~(kStage - 1);   // stages are 64-channel aligned
    ch_hi = rng.y;
  }
  if (ch_lo >= ch_hi) return;
  const int nst = (ch_hi - ch_lo + kStage - 1) / kStage;    // stages per tap
  const int nvalid = min(MB, hd.oblks - orange * MB);       // real output-channel blocks of this row

  // ---- the pixel whose sampling state this lane computes (lanes 32-63 mirror 0-31) ----
  int b, pix;
  bool live;
  if (hd.blocked) {   // tiles in blocked order (hp_common.hpp: hp_wave_segment): the segment never leaves its row
    hp_wave_segment(g, 1, tile, wave, b, pix);
    live = b < g.B;
    pix = live ? pix + (lane & 31) : 0;
    b = live ? b : g.B - 1;
  } else {
    const int n_raw = tile * 128 + wave * 32 + (lane & 31);
    live = n_raw < g.N;
    const int n = live ? n_raw : g.N - 1;
    b = n / g.S_o;
    pix = n - b * g.S_o;
  }
  int oc[ND];
  out_coords<ND>(g, pix, oc);

  const rsrc_t r_xt = make_rsrc(xt, (size_t)g.B * g.S_i * hd.Cp * 2);
  const rsrc_t r_w = make_rsrc(wpf, (size_t)g.K * nchunks * hd.oblks * 1024);

  f32x16 acc[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // with DG > 1 every stage lies inside one deformable group (Cdg is a multiple of 64)
  const int spd = g.DG == 1 ? nst : g.Cdg / 64;   // stages per deformable group
  const int dg0 = g.DG == 1 ? 0 : (ch_lo * 16) / g.Cdg;

  // ---- sampling state: offsets / mask one state ahead, state table in LDS (two slots) ----
  // The RAW 16-bit values are kept until build(): converting them to float inside fetch() is a use of the load
  // right where it is issued -- a full memory round trip per sampling state, exposed (with the conversion in
  // fetch() "state build + fetch" was 22 % / 25 % of the kernel's wave-cycles at cfg5 / cfg3, tools/b1_timing.py).
  Raw dlr[ND], mlr;
  const Raw *off_px = offset + (int64_t)b * g.DG * (ND * g.K) * g.S_o + pix;
  const Raw *msk_px = MOD ? mask + (int64_t)b * g.DG * g.K * g.S_o + pix : nullptr;
  const int px_base = b * g.S_i;
  auto store_state = [&](const TapCoef<ND, float> &tc, float ml, int slot) {
    HpCorners<ND> hc;
    hp_corners<ND>(tc, hc);
    int *sp = St + (slot * 32 + (lane & 31)) * SW;
    int ev[SW];
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) {
      ev[ci] = hc.idx[ci] >= 0 ? (px_base + hc.idx[ci]) * hd.Cp * 2 : kHpOob;
      ev[NC + ci] = __float_as_int(hc.w[ci] * ml);
    }
#pragma unroll
    for (int q = 0; q < SW; q += 4) *reinterpret_cast<int4 *>(sp + q) = make_int4(ev[q], ev[q + 1], ev[q + 2], ev[q + 3]);
  };
  // states of a tap: one per run of `spd` stages (deformable group); state index = tap * ndg + run
  const int ndg = g.DG == 1 ? 1 : (nst + spd - 1) / spd;
  const int nstates = g.K * ndg;
  // the state THIS lane builds next: (b_tap, b_run) with the tap's coordinates kept incrementally (a per-lane tap
  // would otherwise cost two integer divisions per build); lanes 32-63 start one state ahead of lanes 0-31
  int b_tap = 0, b_run = 0, b_tcd[ND];
#pragma unroll
  for (int a = 0; a < ND; ++a) b_tcd[a] = 0;
  auto advance = [&]() {   // one state further
    if (++b_run == ndg) {
      b_run = 0;
      ++b_tap;
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
  if (kh) advance();
  auto fetch_pair = [&]() {   // offsets / mask of this lane's next state (clamped past the last one: built, never read)
    const int tp = min(b_tap, g.K - 1);
    const int idx = (dg0 + b_run) * g.K + tp;
    const Raw *op = off_px + (int64_t)idx * (ND * g.S_o);
#pragma unroll
    for (int a = 0; a < ND; ++a) dlr[a] = op[(int64_t)a * g.S_o];
    if (MOD) mlr = msk_px[(int64_t)idx * g.S_o];
  };
  auto build_pair = [&](int sig0) {   // states sig0 (lanes 0-31) and sig0 + 1 (lanes 32-63) from the fetched values
    float dl[ND], ml = 1.f;
#pragma unroll
    for (int a = 0; a < ND; ++a) dl[a] = T::ldf(&dlr[a]);
    if (MOD) ml = T::ldf(&mlr);
    TapCoef<ND, float> tc;
    make_tap<ND, float>(g, oc, b_tcd, dl, false, tc);
    const int sg = sig0 + kh;
    store_state(tc, ml, sg - (sg / 3) * 3);
    advance();
    advance();
  };

  // ---- weight staging: fragment f of a stage = (chunk f / MB, block f % MB); wave w moves
  // fragments w, w + 4, ...; with groups only the blocks a chunk can reach, compacted ----
  constexpr int FPW = (kStage * MB + 3) / 4;
  U4 wr[FPW];
  auto w_load = [&](int tap, int ch0) {
#pragma unroll
    for (int k = 0; k < FPW; ++k) {
      const int f = wave + 4 * k;
      const int j = f / MB, ob = f % MB;
      const int ch = ch0 + j;
      bool on = f < kStage * MB && ch < ch_hi;
      if (GRP) {
        if (on) {
          const int2 e = ct[ch];
          on = ob >= e.x && ob < e.x + e.y;
        }
      } else {
        on = on && ob < nvalid;
      }
      // unconditional load from an out-of-range offset when the fragment is not needed (returns 0, no
      // traffic): a conditional load into the array put wr[] in scratch (36 bytes per lane at MB = 4)
      wr[k] = buf_load4u(r_w, on ? lane * 16 : kHpOob, on ? ((tap * nchunks + ch) * hd.oblks + orange * MB + ob) * 1024 : 0);
    }
  };
  auto w_store = [&](U4 *Ab, int ch0) {
#pragma unroll
    for (int k = 0; k < FPW; ++k) {
      const int f = wave + 4 * k;
      const int j = f / MB, ob = f % MB;
      const int ch = ch0 + j;
      if (f < kStage * MB && ch < ch_hi) {
        if (GRP) {
          const int2 e = ct[ch];
          if (ob >= e.x && ob < e.x + e.y) Ab[(j * nmax + (ob - e.x)) * 64] = wr[k];
        } else if (ob < nvalid) {
          Ab[(j * MB + ob) * 64] = wr[k];
        }
      }
    }
  };

  // ---- gathers: pixel group pg (8 pixels) of the stage with state slot `sp0`, channel base cbase2 ----
  struct Set { U4 v[NC]; float w[NC]; };
  const int lane_off = oc8 * 16;
  auto issue = [&](Set &s, const int *sp0, int pg, int cbase2) {
    const int *sp = sp0 + (pg * 8) * SW;
    int ev[SW];
#pragma unroll
    for (int q = 0; q < SW; q += 4) {
      const int4 x = *reinterpret_cast<const int4 *>(sp + q);
      ev[q] = x.x; ev[q + 1] = x.y; ev[q + 2] = x.z; ev[q + 3] = x.w;
    }
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) {
      s.v[ci] = buf_load4u(r_xt, ev[ci] + lane_off, cbase2);
      s.w[ci] = __int_as_float(ev[NC + ci]);
    }
  };
  Raw *bt_w = Bt + gp * kBtP + oc8 * 8;                       // this lane's write slot (pixel group 0)
  const Raw *bt_r = Bt + (lane & 31) * kBtP + 8 * kh;         // this lane's B-fragment row
  auto interp = [&](const Set &s, int pg) {
    float col[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) col[j] = 0.f;
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) mac8<T>(col, s.v[ci], s.w[ci]);
    *reinterpret_cast<U4 *>(bt_w + pg * 8 * kBtP) = pack8<T>(col);
  };

  // ---- prologue ----
  int tap = 0, st = 0;                 // stage being processed
  int slot = 0;
  int sig = 0;                         // index of the state being processed
  fetch_pair();
  build_pair(0);
  if (nstates > 2) fetch_pair();       // the pair after that
  w_load(0, ch_lo);
  Set sa, sb;
  const int *st_lane = St + gp * SW;   // this lane's row of pixel group 0, slot 0
  issue(sa, st_lane, 0, ch_lo * 32);
  const int S = g.K * nst;
  for (int s = 0; s < S; ++s) {
    const int ch0 = ch_lo + st * kStage;
    U4 *Ab = As + ((s & 1) * kStage * nmax) * 64 + lane;
    // position of the next stage and whether its sampling state is a new one
    int tap1 = tap, st1 = st + 1;
    if (st1 == nst) { st1 = 0; ++tap1; }
    const bool new_state = st1 == 0 || (g.DG > 1 && st1 % spd == 0);
    int slot_next = slot;
    if (new_state && s + 1 < S) {
      ++sig;
      slot_next = sig - (sig / 3) * 3;
      if ((sig & 1) == 0) {              // states sig and sig + 1 are not built yet
        build_pair(sig);
        if (sig + 2 < nstates) fetch_pair();
      }
    }
    const int *sp_cur = st_lane + slot * 32 * SW;
    // ---- gather + interpolate the 4 pixel groups of this stage; the first group of the next stage
    // is requested before the matrix phase ----
    issue(sb, sp_cur, 1, ch0 * 32);
    interp(sa, 0);
    issue(sa, sp_cur, 2, ch0 * 32);
    interp(sb, 1);
    // the offsets / mask fetched at the top of this stage are OLDER than the gathers interp(sb, 1) has just waited
    // for: naming them as used here costs no wait, and build() of the next state does not have to drain the queue
#pragma unroll
    for (int a = 0; a < ND; ++a) asm volatile("" : "+v"(dlr[a]));
    if (MOD) asm volatile("" : "+v"(mlr));
    issue(sb, sp_cur, 3, ch0 * 32);
    interp(sa, 2);
    if (s + 1 < S) issue(sa, st_lane + slot_next * 32 * SW, 0, (ch_lo + st1 * kStage) * 32);
    interp(sb, 3);
    // ---- weights of this stage -> LDS; next stage's weights requested ----
    w_store(Ab, ch0);
    __syncthreads();
    if (s + 1 < S) w_load(tap1, ch_lo + st1 * kStage);
    // ---- matrix phase ----
    const int nj = min(kStage, ch_hi - ch0);
#pragma unroll
    for (int j = 0; j < kStage; ++j) {
      if (j < nj) {
        const U4 bfrag = *reinterpret_cast<const U4 *>(bt_r + j * 16);
        if (GRP) {
          const int2 e = ct[ch0 + j];
#pragma unroll
          for (int ob = 0; ob < MB; ++ob)
            if (ob >= e.x && ob < e.x + e.y) acc[ob] = T::mfma(Ab[(j * nmax + (ob - e.x)) * 64], bfrag, acc[ob]);
        } else {
#pragma unroll
          for (int ob = 0; ob < MB; ++ob)
            if (ob < nvalid) acc[ob] = T::mfma(Ab[(j * MB + ob) * 64], bfrag, acc[ob]);
        }
      }
    }
    slot = slot_next;
    tap = tap1;
    st = st1;
  }

  // ---- epilogue: + bias, store [B, O, S_o]; lanes 0-31 -> 32 consecutive pixels ----
  if (live) {
    // the lane's 16 MB bias values first, as independent loads (inside the store loop below each one sat between two dependent
    // stores: +0.09 ms on the cfg5 forward with bias, profiles/r05_experiments.md 28)
    if (g.with_bias) {
#pragma unroll
      for (int ob = 0; ob < MB; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = (orange * MB + ob) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
          acc[ob][r] += T::ldf(bias + (o < g.O ? o : 0));
        }
    }
#pragma unroll
    for (int ob = 0; ob < MB; ++ob)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = (orange * MB + ob) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (o < g.O) T::stf(output + ((int64_t)b * g.O + o) * g.S_o + pix, acc[ob][r]);
      }
  }
}

}  // namespace


size_t hp_fwd2_lds_bytes(const Geom &g, const HpDims &hd) {
  const int nc = 1 << g.nd;
  const size_t win = 0;
  return (size_t)2 * kStage * (g.G == 1 ? hd.MB : hd.fwd_nmax) * 1024 + (size_t)4 * 32 * kBtP * 2 + (size_t)4 * kStSlots * 32 * 2 * nc * 4 + win;
}

template <int ND, bool MOD, typename T>
static int launch_fwd2_hp(const Geom &g, const HpDims &hd, const Tensors &t, const void *xt,
                          const void *wpf, const int2 *ctab, hipStream_t stream) {
  using Raw = typename T::Raw;
  const dim3 grid((g.N + 127) / 128, hd.oranges);
  const size_t lds = hp_fwd2_lds_bytes(g, hd);
#define HP_FWD2(MBV)                                                                             \
  do {                                                                                           \
    if (g.G == 1) HP_FWD2_(MBV, false); else HP_FWD2_(MBV, true);                                \
  } while (0)
#define HP_FWD2_(MBV, GRPV)                                                                      \
  do {                                                                                           \
    if (lds > 64 * 1024) {                                                                       \
      hipError_t ea = hipFuncSetAttribute((const void *)hp_fwd2_kernel<ND, MOD, T, MBV, GRPV>,    \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (ea != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(ea)); return MDCONV_ELAUNCH; } \
    }                                                                                            \
    hp_debug_plan("hp_fwd2", hp_fwd2_kernel<ND, MOD, T, MBV, GRPV>, 256, lds, (long)grid.x * grid.y);   \
    hipLaunchKernelGGL((hp_fwd2_kernel<ND, MOD, T, MBV, GRPV>), grid, dim3(256), lds, stream, g, hd, \
                       (const Raw *)xt, (const U4 *)wpf, (const Raw *)t.bias, (const Raw *)t.offset, \
                       (const Raw *)t.mask, (Raw *)t.output, ctab);                              \
  } while (0)
  switch (hd.MB) {
    case 1: HP_FWD2(1); break;
    case 2: HP_FWD2(2); break;
    default: HP_FWD2(4); break;   // (hp_dims: at most 4 output-channel blocks per row)
  }
#undef HP_FWD2
#undef HP_FWD2_
  return check_launch("hp_fwd2");
}

int hp_forward2_launch(const Geom &g, const HpDims &hd, int dtype, const Tensors &t, const void *xt,
                       const void *wpf, const int2 *ctab, hipStream_t stream) {
#define HP_DISPATCH(T)                                                                       \
  do {                                                                                       \
    if (g.nd == 2)                                                                           \
      return g.modulated ? launch_fwd2_hp<2, true, T>(g, hd, t, xt, wpf, ctab, stream)        \
                         : launch_fwd2_hp<2, false, T>(g, hd, t, xt, wpf, ctab, stream);      \
    return g.modulated ? launch_fwd2_hp<3, true, T>(g, hd, t, xt, wpf, ctab, stream)          \
                       : launch_fwd2_hp<3, false, T>(g, hd, t, xt, wpf, ctab, stream);        \
  } while (0)
  if (dtype == MDCONV_F16) HP_DISPATCH(F16);
  HP_DISPATCH(BF16);
#undef HP_DISPATCH
}

}  // namespace mdconv
