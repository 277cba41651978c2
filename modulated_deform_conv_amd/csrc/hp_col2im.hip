// hp_col2im.hip -- grad_input of the native 16-bit path: inverse scatter map + gather.
//
// Reference: the 4 / 8 atomics per sample of the gradient kernels (mdeformable_conv.cu:282-293,
// mdeformable_conv3d.cu:341-379).  Here the data-dependent scatter is inverted once per call
// (count [inside the fused backward kernel] -> scan -> fill; integer atomics only) into lists keyed
// by (image, deformable group, extended anchor): ONE entry per SAMPLE (tap, output pixel), keyed by
// its low corner (hp_common.hpp: SampleAnchor) -- scattered global atomics run at 27 G/s on this
// chip whatever their scope (tools/ubench_atomic.hip), so their NUMBER is what the build costs, and
// keying by sample instead of by corner pair divides it by 2^(ND-1).  An entry carries
// (tap * S_o + output pixel, weight * mask on columns cl and cl + 1, low / high weights of the
// outer axes).  The gather walks target pixels along the last axis: a group of LPD lanes (8
// channels each) visits, per target, the 2^(ND-1) anchor rows that can reach it, reads every 16-bit
// grad_col row [b][tap][pix][c] once per (entry, row) with 16-byte loads, accumulates in fp32
// (`cur` for the column, `nxt` for column + 1) and writes grad_input [B, C, S_i] through an LDS
// transpose.  No floating-point atomics.
#include "hp_kernels.hpp"

namespace mdconv {

namespace {

int grid_for(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

__global__ __launch_bounds__(256) void hp_zero_int_kernel(int *__restrict__ p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0;
}

// exclusive scan of cnt[seg][0..S) -> rowptr[seg][0..S]: grid (chunks of kScanChunk elements,
// segments).  A workgroup first sums everything before its chunk (the counters are L2-resident and
// that is at most S reads), then scans its chunk -- one workgroup per SEGMENT, as in round 1, left
// 8 workgroups walking 35 k anchors each at the 3-D shards (0.12 / 0.22 ms at cfg4 / cfg5).
__global__ __launch_bounds__(256) void hp_csr_scan_kernel(int S, const int *__restrict__ cnt,
                                            int *__restrict__ rowptr) {
  csr_scan_chunk(S, cnt, rowptr);   // mdconv_common.hpp
}

// Long entry = 2 x int4, fp32 fields: (src, wx, wy, rl0), (rh0, rl1, rh1, anchor)  [2-D: rl1 = rl0, rh1 = rh0].
// Short entry = 1 x int4 (round 4: half the list bytes; cfg3 writes 3.6 M of them), 2-D fp16 tensors only:
// (src, (rh0 wx, rh0 wy), (rl0 wx, rl0 wy), anchor) -- the four weight x mask products as two packed fp16 pairs: 11
// significant bits, what the fp16 grad_col rows they multiply carry, and the mask of an fp16 call cannot leave the fp16
// range.  bf16 tensors keep LONG entries in 2-D too (round 5): fp16 pairs would lose a bf16 mask above 65504 or a
// product below 6e-8 (advisor, round 4), bf16 pairs would round every weight to 8 bits -- a second rounding per term
// that `test_bf16_two_pass_gather_rounds_once_like_the_one_pass_gather` (and its 2-D sibling) forbid.
template <int ND, typename T> struct ShortEntry { static constexpr bool value = false; };
template <> struct ShortEntry<2, F16> { static constexpr bool value = true; };
template <int ND, bool MOD, typename T>
__global__ __launch_bounds__(256) void hp_csr_fill_kernel(Geom g, int S_e,
                                                          const typename T::Raw *__restrict__ offset,
                                                          const typename T::Raw *__restrict__ mask,
                                                          int *__restrict__ cursor,
                                                          const int *__restrict__ rowptr,
                                                          int4 *__restrict__ entries) {
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
    for (int a = 0; a < ND; ++a) delta[a] = T::ldf(offset + ob + (int64_t)a * g.S_o);
    TapCoef<ND, float> tc;
    make_tap<ND, float>(g, oc, tcd, delta, true, tc);
    const float m = MOD ? T::ldf(mask + ((int64_t)seg * g.K + tap) * g.S_o + pix) : 1.f;
    SampleAnchor<ND> sa;
    sample_anchor<ND>(g, tc, m, sa);
    if (sa.on) {
      // the counters double as cursors, counted down: no clearing pass between scan and fill
      const int pos = rowptr[(int64_t)seg * (S_e + 1) + sa.qa] + atomicSub(cursor + (int64_t)seg * S_e + sa.qa, 1) - 1;
      if constexpr (ShortEntry<ND, T>::value) {
        entries[(int64_t)seg * ((int64_t)g.K * g.S_o) + pos] =
            make_int4(tap * g.S_o + pix, (int)T::pack(sa.rh[0] * sa.wx, sa.rh[0] * sa.wy),
                      (int)T::pack(sa.rl[0] * sa.wx, sa.rl[0] * sa.wy), sa.qa);
      } else {
        int4 *e = entries + ((int64_t)seg * ((int64_t)g.K * g.S_o) + pos) * 2;
        e[0] = make_int4(tap * g.S_o + pix, __float_as_int(sa.wx), __float_as_int(sa.wy), __float_as_int(sa.rl[0]));
        e[1] = make_int4(__float_as_int(sa.rh[0]), __float_as_int(sa.rl[ND - 2]), __float_as_int(sa.rh[ND - 2]), sa.qa);
      }
    }
  }
}

constexpr int kRun = 8;   // targets per run

// LPD lanes (8 channels each) follow one list; a wave walks 64 / LPD runs of kRun consecutive
// targets side by side; workgroup tile = 4 * (64 / LPD) runs.  Channel units of LPD * 8 channels
// (one deformable group each when DG > 1) are processed one after the other.
template <int ND, typename T, int LPD>
__global__ __launch_bounds__(256) void hp_col2im_kernel(Geom g, HpDims hd, int S_e,
                                                        const typename T::Raw *__restrict__ gcol,
                                                        const int *__restrict__ rowptr,
                                                        const int4 *__restrict__ entries,
                                                        typename T::Raw *__restrict__ grad_input) {
  using Raw = typename T::Raw;
  constexpr int L = ND - 1, NR = 1 << L;   // anchor rows that reach a target
  constexpr int NQ = 64 / LPD, RUNS = 4 * NQ, QT = RUNS * kRun;
  constexpr int CW = LPD * 8;              // channels per unit
  constexpr int UB = LPD < 8 ? LPD : 8;    // row loads in flight per step
  constexpr int TP = QT + 2;               // LDS pitch (16-bit elements)
  __shared__ Raw tile[CW * TP];
  const int qtiles = (g.S_i + QT - 1) / QT;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int b = bid / qtiles;
  const int q0 = (bid - b * qtiles) * QT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane / LPD, r = lane % LPD;
  const rsrc_t r_gc = make_rsrc(gcol + (size_t)b * g.K * g.S_o * hd.Cp, (size_t)g.K * g.S_o * hd.Cp * 2);
  const int cseg = g.DG == 1 ? hd.Cp : g.Cdg;   // channels that share one list
  const int upd = (cseg + CW - 1) / CW;         // units per segment
  const int units = g.DG * upd;
  const int qs = q0 + (wave * NQ + j) * kRun;
  const int W = g.in_sz[L], H1 = ND == 3 ? g.in_sz[1] + 1 : 1;
  for (int u = 0; u < units; ++u) {
    const int dg = u / upd;
    const int c_lo = dg * cseg + (u - dg * upd) * CW;          // first channel of the unit
    const int c_end = min(dg * cseg + cseg, hd.Cp);            // end of the segment's channels
    const int c8 = c_lo + r * 8;
    const bool chan_on = c8 < c_end;
    const int seg = b * g.DG + dg;
    const int *rp = rowptr + (int64_t)seg * (S_e + 1);
    const int4 *ent = entries + (int64_t)seg * ((int64_t)g.K * g.S_o) * (ShortEntry<ND, T>::value ? 1 : 2);
    const int c_voff = chan_on ? c8 * 2 : kHpOob;
    // coordinates of the target of the current step (first step: qs - 1, the carry-in column)
    int tc[ND];
    {
      const int a0 = max(qs - 1, 0);
      int rem = a0;
#pragma unroll
      for (int a = L; a > 0; --a) { tc[a] = rem % g.in_sz[a]; rem /= g.in_sz[a]; }
      tc[0] = rem;
      if (qs - 1 < 0) tc[L] = -1;
    }
    float cur[8], nxt[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) cur[k] = nxt[k] = 0.f;
    for (int step = 0; step <= kRun; ++step) {
      const int a = qs - 1 + step;
      const bool on = a >= 0 && a < g.S_i;
#pragma unroll
      for (int s = 0; s < NR; ++s) {
        // anchor row s: extended low index = target + s_a on every outer axis
        int er = 0;
#pragma unroll
        for (int ax = 0; ax < L; ++ax) er = er * (g.in_sz[ax] + 1) + tc[ax] + ((s >> (L - 1 - ax)) & 1);
        const int ea = er * W + tc[L];
        const int e0 = on ? rp[ea] : 0, e1 = on ? rp[ea + 1] : 0;
        for (int base = e0; __any(base < e1); base += LPD) {
          const int cnt = max(0, min(LPD, e1 - base));
          int src_m = 0;
          float fx_m = 0.f, fy_m = 0.f;   // weights 0, row 0 beyond the list
          if (r < cnt) {
            // target = low + 1 - s_a on axis a: s_a = 1 -> the low side (rl), 0 -> the high side (rh)
            if constexpr (ShortEntry<ND, T>::value) {
              const int4 e4 = ent[base + r];
              const u32 pr = (u32)(s ? e4.z : e4.y);
              src_m = e4.x;
              fx_m = T::lo(pr);
              fy_m = T::hi(pr);
            } else {
              const int4 ea4 = ent[(int64_t)(base + r) * 2], eb4 = ent[(int64_t)(base + r) * 2 + 1];
              float rw = ((s >> (L - 1)) & 1) ? __int_as_float(ea4.w) : __int_as_float(eb4.x);
              if (ND == 3) rw *= (s & 1) ? __int_as_float(eb4.y) : __int_as_float(eb4.z);
              src_m = ea4.x;
              fx_m = rw * __int_as_float(ea4.y);
              fy_m = rw * __int_as_float(ea4.z);
            }
          }
#pragma unroll
          for (int u0 = 0; u0 < LPD; u0 += UB) {
            U4 v[UB];
            float wx[UB], wy[UB];
#pragma unroll
            for (int k = 0; k < UB; ++k) {
              const int src = __shfl(src_m, u0 + k, LPD);
              wx[k] = __shfl(fx_m, u0 + k, LPD);
              wy[k] = __shfl(fy_m, u0 + k, LPD);
              v[k] = buf_load4u(r_gc, src * hd.Cp * 2 + c_voff, 0);
            }
#pragma unroll
            for (int k = 0; k < UB; ++k) {
              mac8<T>(cur, v[k], wx[k]);
              mac8<T>(nxt, v[k], wy[k]);
            }
          }
        }
      }
      if (step > 0 && chan_on) {
        Raw *tp = tile + (r * 8) * TP + (a - q0);
#pragma unroll
        for (int k = 0; k < 8; ++k) T::stf(tp + k * TP, cur[k]);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) { cur[k] = nxt[k]; nxt[k] = 0.f; }
      // next target along the flattened image
      if (++tc[L] == W) {
        tc[L] = 0;
        if (ND == 3) { if (++tc[1] == g.in_sz[1]) { tc[1] = 0; ++tc[0]; } }
        else ++tc[0];
      }
    }
    __syncthreads();
    // transpose out: consecutive threads -> consecutive q of one channel
    for (int x = threadIdx.x; x < CW * QT; x += 256) {
      const int cl = x / QT, ql = x - cl * QT;
      const int c = c_lo + cl, q = q0 + ql;
      if (c < min(c_end, g.C) && q < g.S_i) {
        Raw *dst = grad_input + ((int64_t)b * g.C + c) * g.S_i + q;
        const float v = tile[cl * TP + ql];
        T::stf(dst, g.acc_data ? T::ldf(dst) + v : v);
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// Two-pass gather (round 3): every grad_col row and every list entry is read ONCE.
// The one-pass kernel above visits, per target, the 2^(ND-1) anchor rows that reach it, so a sample's
// row is fetched 2^(ND-1) times -- and at the 3-D shards those repeats are 32 workgroups apart: 14 GB of
// HBM reads for 3.6 GB of rows at cfg5 (L2 hit 14 %, 7 TB/s: HBM-bound, profiles/r02_hp_counters.md).
// Pass 1 walks ANCHOR rows instead and keeps, per anchor column, one partial sum for each of the
// 2^(ND-1) target rows the anchor row feeds (s: bit a set = the target one lower on outer axis a, i.e.
// weight rl_a; clear = weight rh_a) -- the same multiply-adds as before, spread over NS accumulators --
// and writes them as 16-bit rows A[segment][anchor][s][channels].  Pass 2 is a 2^(ND-1)-point stencil
// over A (target t takes s from anchor row t + s) plus the transpose to [B, C, S_i].
// ---------------------------------------------------------------------------------------------
constexpr int kRunA = 16;   // anchors per run: the carry-in anchor is read twice (1 / 16 of the rows)

// Storage of the partial sums: the tensors' own 16-bit type for fp16 (11 significant bits; two to four of them are
// added per target), fp32 for bf16 -- 8 bits per partial sum and then cancellation between them was a real loss on
// grad_input (advisor, round 3), so bf16 rounds ONCE, at the grad_input store, like the one-pass kernel.
template <typename T> struct SumStore { using type = typename T::Raw; };
template <> struct SumStore<BF16> { using type = float; };

template <int ND, typename T, int LPD>
__global__ __launch_bounds__(256) void hp_col2im_sums_kernel(Geom g, HpDims hd, int S_e,
                                                             const typename T::Raw *__restrict__ gcol,
                                                             const int *__restrict__ rowptr,
                                                             const int4 *__restrict__ entries,
                                                             typename SumStore<T>::type *__restrict__ sums) {
  using Sum = typename SumStore<T>::type;
  constexpr bool WIDE = sizeof(Sum) == 4;
  constexpr int L = ND - 1, NS = 1 << L;
  constexpr int NQ = 64 / LPD, RUNS = 4 * NQ;
  constexpr int UB = 4;            // rows per load group; two groups are in flight
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane / LPD, r = lane % LPD;
  const int cseg = g.DG == 1 ? hd.Cp : g.Cdg;   // channels that share one list
  const int runs_per_seg = (S_e + kRunA - 1) / kRunA;
  const int blocks_per_seg = (runs_per_seg + RUNS - 1) / RUNS;
  const int seg = blockIdx.x / blocks_per_seg;                  // b * DG + dg
  const int run = (blockIdx.x - seg * blocks_per_seg) * RUNS + wave * NQ + j;
  const int b = seg / g.DG, dg = seg - b * g.DG;
  const bool run_on = run * kRunA < S_e;
  const int a_lo = run_on ? run * kRunA : 0;                    // first anchor this run writes
  const int a_last = run_on ? min(a_lo + kRunA, S_e) - 1 : -1;  // last one
  const rsrc_t r_gc = make_rsrc(gcol + (size_t)b * g.K * g.S_o * hd.Cp, (size_t)g.K * g.S_o * hd.Cp * 2);
  const int *rp = rowptr + (int64_t)seg * (S_e + 1);
  const int4 *ent = entries + (int64_t)seg * ((int64_t)g.K * g.S_o) * (ShortEntry<ND, T>::value ? 1 : 2);
  const bool chan_on = r * 8 < cseg;
  const int c_voff = chan_on ? (dg * cseg + r * 8) * 2 : kHpOob;
  Sum *out = sums + ((int64_t)seg * S_e * NS) * cseg + r * 8;
  // The lists of consecutive anchors are contiguous in `entries`, so a run streams ONE entry range --
  // from the carry-in anchor a_lo - 1 (only its column + 1 part lands in this run) to a_last -- in
  // batches of LPD entries, rows loaded UB at a time with two groups in flight; an entry names its
  // anchor, and the accumulators are flushed whenever the anchor advances (empty anchors included).
  // No per-anchor rowptr -> entry -> row dependency chain is left.
  const int e_end = run_on ? rp[a_last + 1] : 0;
  int e_pos = run_on ? rp[max(a_lo - 1, 0)] : 0;
  int cur_a = a_lo - 1;
  float cur[NS][8], nxt[NS][8];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int k = 0; k < 8; ++k) cur[s][k] = nxt[s][k] = 0.f;
  auto flush = [&]() {   // anchor cur_a is complete: write its sums (not for the carry-in anchor), shift the column carry
    if (cur_a >= a_lo && chan_on) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        Sum *o = out + ((int64_t)cur_a * NS + s) * cseg;
        if constexpr (WIDE) {
          reinterpret_cast<float4 *>(o)[0] = make_float4(cur[s][0], cur[s][1], cur[s][2], cur[s][3]);
          reinterpret_cast<float4 *>(o)[1] = make_float4(cur[s][4], cur[s][5], cur[s][6], cur[s][7]);
        } else {
          *reinterpret_cast<U4 *>(o) = pack8<T>(cur[s]);
        }
      }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int k = 0; k < 8; ++k) { cur[s][k] = nxt[s][k]; nxt[s][k] = 0.f; }
    ++cur_a;
  };
  while (__any(e_pos < e_end)) {
    // this lane's entry of the batch
    // this lane's entry: row, anchor and the weight of the row towards (target row s, column x / x + 1)
    int src_m = 0, anc_m = 0x7fffffff;
    float px_m[NS], py_m[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) px_m[s] = py_m[s] = 0.f;
    if (e_pos + r < e_end) {
      if constexpr (ShortEntry<ND, T>::value) {
        const int4 e4 = ent[e_pos + r];
        src_m = e4.x;
        px_m[0] = T::lo((u32)e4.y); py_m[0] = T::hi((u32)e4.y);
        px_m[1] = T::lo((u32)e4.z); py_m[1] = T::hi((u32)e4.z);
        anc_m = e4.w;
      } else {
        const int4 ea4 = ent[(int64_t)(e_pos + r) * 2], eb4 = ent[(int64_t)(e_pos + r) * 2 + 1];
        src_m = ea4.x;
        const float wx = __int_as_float(ea4.y), wy = __int_as_float(ea4.z);
        const float f0l = __int_as_float(ea4.w), f0h = __int_as_float(eb4.x);
        const float f1l = __int_as_float(eb4.y), f1h = __int_as_float(eb4.z);
        // s: bit a set = the target one lower on outer axis a (weight rl_a), clear = rh_a; 2-D has one outer axis
        const float fa[4] = {ND == 3 ? f0h * f1h : f0h, ND == 3 ? f0h * f1l : f0l, f0l * f1h, f0l * f1l};
#pragma unroll
        for (int s = 0; s < NS; ++s) { px_m[s] = fa[s] * wx; py_m[s] = fa[s] * wy; }
        anc_m = eb4.w;
      }
    }
    const int cnt = max(0, min(LPD, e_end - e_pos));
    U4 va[UB], vb[UB];
    auto load_group = [&](U4 (&v)[UB], int u0) {
#pragma unroll
      for (int k = 0; k < UB; ++k) {
        const int src = __shfl(src_m, u0 + k, LPD);
        v[k] = buf_load4u_nt(r_gc, src * hd.Cp * 2 + c_voff, 0);
      }
    };
    auto use_group = [&](const U4 (&v)[UB], int u0) {
#pragma unroll
      for (int k = 0; k < UB; ++k) {
        const int anc = __shfl(anc_m, u0 + k, LPD);
        float px[NS], py[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) { px[s] = __shfl(px_m[s], u0 + k, LPD); py[s] = __shfl(py_m[s], u0 + k, LPD); }
        if (u0 + k < cnt) {
          while (cur_a < anc) flush();
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            mac8<T>(cur[s], v[k], px[s]);
            mac8<T>(nxt[s], v[k], py[s]);
          }
        }
      }
    };
    static_assert(LPD % (2 * UB) == 0 || LPD == UB, "batch = an even number of load groups");
    load_group(va, 0);
#pragma unroll 1
    for (int u0 = 0; u0 < LPD; u0 += 2 * UB) {
      if (!__any(u0 < cnt)) break;          // (wave-uniform) nothing left in the batch for any run of the wave
      if (LPD > UB) load_group(vb, u0 + UB);
      use_group(va, u0);
      if (LPD > UB) {
        if (u0 + 2 * UB < LPD) load_group(va, u0 + 2 * UB);
        use_group(vb, u0 + UB);
      }
    }
    e_pos += LPD;
  }
  while (cur_a <= a_last) flush();
}

// ---------------------------------------------------------------------------------------------
// Pass 1 on the matrix cores (round 5).  The partial sums of an anchor window are a small GEMM:
//     A[(anchor, s)][c] = sum over the window's list entries e of  Wt[(anchor, s)][e] * grad_col[src(e)][c]
// with Wt sparse -- entry e of anchor a carries px[s] for row (a, s) and py[s] for row (a + 1, s) (the column + 1
// carry of the VALU kernel above).  M = 32 rows = AW anchors x NS target rows (AW = 16 in 2-D, 8 in 3-D), K = 16
// entries per step, N = 32 channels per accumulator block: v_mfma_f32_32x32x16 does in one instruction per 16
// entries and 32 channels what the VALU kernel does with 2 NS x 8 fp32 FMAs per entry and lane, and the per-anchor
// flush (half of that kernel's instructions at cfg3) becomes ONE contiguous 32-row store per window.
//   * a wave owns a run of kRunM consecutive anchors = kRunM / AW windows; per window it streams the entries of
//     anchors [w0 - 1, w0 + AW) (the carry-in anchor's entries were read by the previous window a moment ago: an L2 hit);
//   * per step: lanes 0-15 fetch one entry each (entries run two steps ahead, rows one step ahead), turn it
//     into 2 NS weights in the tensors' 16-bit type and drop them into a zeroed wave-private LDS tile Wt[16][40] at
//     column 4 + NS (anchor - w0) (a margin of 4 columns takes the carry-in anchor, the right margin the carry-out);
//     all lanes fetch the 16 grad_col rows (cseg / 8 lanes per row, 16 bytes each) into the tile R[16][cseg + 32];
//     both operands come out of LDS with transposing reads (K-major fragments of row-major tiles, lds_tr2);
//   * no workgroup barrier anywhere: the tiles are wave-private, a wave reads what it wrote.
// fp32 accumulation; the weights carry 11 (fp16) / 8 (bf16) significant bits like the rows they multiply.
constexpr int kRunM = 32;    // anchors per wave
constexpr int kWtP = 40;     // pitch (16-bit elements) of the weight tile: 4 margin + 32 + up to 4 carry-out columns
constexpr int kWtM = 4;      // left margin

// HALF (round 6): lists of 16 channels (deformable groups of 16 channels) -- half a 32-column block: the rows fill columns
// 0-15 of the row tile, columns 16-31 hold whatever the tile held (a column of the B operand reaches only its own column of
// the product, and those are never stored).
template <int ND, typename T, int NB, bool HALF = false>
__global__ __launch_bounds__(256, NB <= 2 ? 4 : (NB == 4 ? 3 : 1)) void hp_col2im_sums_mfma_kernel(Geom g, HpDims hd, int S_e,
                                                                  const typename T::Raw *__restrict__ gcol,
                                                                  const int *__restrict__ rowptr,
                                                                  const int4 *__restrict__ entries,
                                                                  typename SumStore<T>::type *__restrict__ sums) {
  using Raw = typename T::Raw;
  using Sum = typename SumStore<T>::type;
  constexpr bool WIDE = sizeof(Sum) == 4;
  // fp16 tensors only: the weights enter the matrix core in the tensors' type, and 8-bit bf16 weights would be a second
  // rounding per term (bf16 tensors keep the fp32-weight VALU kernel above)
  static_assert(ShortEntry<2, T>::value, "fp16 tensors only");
  constexpr int L = ND - 1, NS = 1 << L, AW = 32 / NS;
  static_assert(!HALF || NB == 1, "half blocks: one block");
  constexpr int CS = HALF ? 16 : NB * 32;  // channels that share one list
  constexpr int PB = CS + 32;              // pitch of the row tile
  constexpr int LPR = CS / 8;              // lanes per grad_col row
  constexpr int RPI = 64 / LPR < 16 ? 64 / LPR : 16;   // rows per wave-load (NB = 8: 2; HALF: 16 rows on lanes 0-31)
  constexpr int NLD = 16 / RPI;            // wave-loads per step
  constexpr int RF = WIDE ? 8 : 16;        // rows of the window per flush pass (RF * CS * sizeof(Sum) = 32 CS bytes)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, kh = lane >> 5, pl = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  Raw *Wt = reinterpret_cast<Raw *>(smem + wave * (16 * kWtP * 2 + 16 * PB * 2));   // [16][kWtP]
  Raw *Rt = Wt + 16 * kWtP;                                                          // [16][PB]; flush staging too
  const int runs_per_seg = (S_e + kRunM - 1) / kRunM;
  const int run_id = blockIdx.x * 4 + wave;
  const int seg = run_id / runs_per_seg;                        // b * DG + dg
  if (seg >= g.B * g.DG) return;
  const int a_run = (run_id - seg * runs_per_seg) * kRunM;
  const int b = seg / g.DG, dg = seg - b * g.DG;
  const rsrc_t r_gc = make_rsrc(gcol + (size_t)b * g.K * g.S_o * hd.Cp, (size_t)g.K * g.S_o * hd.Cp * 2);
  const int *rp = rowptr + (int64_t)seg * (S_e + 1);
  const int4 *ent = entries + (int64_t)seg * ((int64_t)g.K * g.S_o) * (ShortEntry<ND, T>::value ? 1 : 2);
  Sum *out = sums + ((int64_t)seg * S_e * NS) * CS;
  // this lane's piece of a grad_col row in the row role: row (lane / LPR) of a wave-load, 16 bytes at channel 8 (lane % LPR)
  const int r_row = lane / LPR, r_piece = lane % LPR;
  const int c_voff = (dg * CS + r_piece * 8) * 2;
  // zero the weight tile once; every step clears what it wrote
  for (int i = lane; i < 16 * kWtP / 8; i += 64) reinterpret_cast<U4 *>(Wt)[i] = U4{0, 0, 0, 0};

  // fragment addresses (hp_gemm2.hip): lane i of a 16-lane group addresses row (i >> 2), element quad (i & 3) of its 4 x 16 block
  const int fr_row = 8 * kh + ((lane & 15) >> 2), fr_col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  for (int w0 = a_run; w0 < min(a_run + kRunM, S_e); w0 += AW) {
    const int e_lo = rp[max(w0 - 1, 0)], e_hi = rp[min(w0 + AW, S_e)];
    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    // Software pipeline over the steps of the window (16 entries each): the entries run TWO steps ahead of the
    // matrix work and the grad_col rows ONE step ahead -- a step's rows are requested while the previous step is
    // multiplied, so the entry -> row dependency costs one exposed round trip per window, not per step
    struct Ent { int4 a, b; };
    auto fetch_entry = [&](int e0) {
      Ent en = {make_int4(-1, 0, 0, 0), make_int4(0, 0, 0, 0)};   // beyond the list: row -1 (parked), no weights
      const int e = e0 + lane;
      if (lane < 16 && e < e_hi) {
        if constexpr (ND == 2) en.a = ent[e];
        else { en.a = ent[(int64_t)e * 2]; en.b = ent[(int64_t)e * 2 + 1]; }
      }
      return en;
    };
    U4 rows_cur[NLD], rows_nxt[NLD];
    auto request_rows = [&](U4 (&rows)[NLD], const Ent &en) {   // entries beyond the list: parked out of range, zeros
      const int src = lane < 16 ? en.a.x : -1;
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int sr = __shfl(src, i * RPI + r_row, 64);
        rows[i] = buf_load4u_nt(r_gc, sr >= 0 ? sr * hd.Cp * 2 + c_voff : kHpOob, 0);
      }
    };
    Ent en0 = fetch_entry(e_lo), en1 = fetch_entry(e_lo + 16);
    if (e_lo < e_hi) request_rows(rows_cur, en0);
    for (int e0 = e_lo; e0 < e_hi; e0 += 16) {
      // ---- weights of this step's entries -> Wt ----
      const bool e_on = lane < 16 && en0.a.x >= 0;
      int wcol = 0;                                    // column of this entry's first weight
      if (e_on) {
        const int anc = ND == 2 ? en0.a.w : en0.b.w;
        wcol = kWtM + NS * (anc - w0);                 // anc in [w0 - 1, w0 + AW): columns [kWtM - NS, kWtM + 32)
        Raw *wp_ = Wt + lane * kWtP + wcol;
        if constexpr (ND == 2) {
          // (rh wx, rh wy), (rl wx, rl wy) -> [px0, px1, py0, py1] = [rh wx, rl wx, rh wy, rl wy]
          const u32 p0 = (u32)en0.a.y, p1 = (u32)en0.a.z;
          reinterpret_cast<u32 *>(wp_)[0] = (p0 & 0xffffu) | (p1 << 16);
          reinterpret_cast<u32 *>(wp_)[1] = (p0 >> 16) | (p1 & 0xffff0000u);
        } else {
          const float wx = __int_as_float(en0.a.y), wy = __int_as_float(en0.a.z);
          const float f0l = __int_as_float(en0.a.w), f0h = __int_as_float(en0.b.x);
          const float f1l = __int_as_float(en0.b.y), f1h = __int_as_float(en0.b.z);
          const float fa0 = f0h * f1h, fa1 = f0h * f1l, fa2 = f0l * f1h, fa3 = f0l * f1l;
          uint2 lo, hi;
          lo.x = T::pack(fa0 * wx, fa1 * wx); lo.y = T::pack(fa2 * wx, fa3 * wx);
          hi.x = T::pack(fa0 * wy, fa1 * wy); hi.y = T::pack(fa2 * wy, fa3 * wy);
          reinterpret_cast<uint2 *>(wp_)[0] = lo;
          reinterpret_cast<uint2 *>(wp_)[1] = hi;
        }
      }
      // ---- next step's rows and the entries after that: in flight behind this step's rows ----
      const bool more = e0 + 16 < e_hi;
      if (more) request_rows(rows_nxt, en1);
      const Ent en2 = fetch_entry(e0 + 32);
      // ---- this step's 16 grad_col rows -> Rt ----
#pragma unroll
      for (int i = 0; i < NLD; ++i)
        if (!HALF || r_row < 16) *reinterpret_cast<U4 *>(Rt + (i * RPI + r_row) * PB + r_piece * 8) = rows_cur[i];
      // ---- 16 entries x 32 NB channels into the window's accumulators ----
      // (compiler fences: the tiles are written and read through differently typed pointers)
      asm volatile("" ::: "memory");
      U4 af;
      lds_tr2(Wt + fr_row * kWtP + kWtM + fr_col, 4 * kWtP, af);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        U4 bf;
        lds_tr2(Rt + fr_row * PB + nb * 32 + fr_col, 4 * PB, bf);
        acc[nb] = T::mfma(af, bf, acc[nb]);
      }
      asm volatile("" ::: "memory");
      // clear this step's weights (the tile stays zero outside the step's 16 x 2 NS values)
      if (e_on) {
        Raw *wp_ = Wt + lane * kWtP + wcol;
        if constexpr (ND == 2) {
          reinterpret_cast<u32 *>(wp_)[0] = 0u;
          reinterpret_cast<u32 *>(wp_)[1] = 0u;
        } else {
          reinterpret_cast<uint2 *>(wp_)[0] = make_uint2(0u, 0u);
          reinterpret_cast<uint2 *>(wp_)[1] = make_uint2(0u, 0u);
        }
      }
      en0 = en1;
      en1 = en2;
#pragma unroll
      for (int i = 0; i < NLD; ++i) rows_cur[i] = rows_nxt[i];
    }
    // ---- flush: rows (anchor - w0) NS + s, contiguous in `sums`; RF rows per pass through the row tile ----
    Sum *Ft = reinterpret_cast<Sum *>(Rt);   // [RF][CS]
#pragma unroll
    for (int ps = 0; ps < 32 / RF; ++ps) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          // accumulator r of this lane = row (r & 3) + 8 (r >> 2) + 4 kh, column nb * 32 + pl
          if ((8 * (r >> 2)) / RF == ps) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kh - ps * RF;
            if (HALF && pl >= CS) continue;
            if constexpr (WIDE) Ft[row * CS + nb * 32 + pl] = acc[nb][r];
            else T::stf(reinterpret_cast<Raw *>(Ft) + row * CS + nb * 32 + pl, acc[nb][r]);
          }
        }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int q = lane + 64 * i;                          // 16-byte piece of the pass: 2 CS pieces in all
        if (HALF && q >= 2 * CS) continue;
        const int row = (q * 16 / (int)sizeof(Sum)) / CS + ps * RF;
        const U4 v = reinterpret_cast<const U4 *>(Ft)[q];
        if (w0 + row / NS < S_e)
          *reinterpret_cast<U4 *>(reinterpret_cast<unsigned char *>(out + (int64_t)w0 * NS * CS) + (size_t)ps * RF * CS * sizeof(Sum) + (size_t)q * 16) = v;
      }
      asm volatile("" ::: "memory");
    }
  }
}

// pass 2: grad_input[b][c][t] (+)= sum_s A[segment(b, c)][anchor row t + s][x][s][c]; workgroup = 64
// consecutive targets x 64 channels, lanes = (target, channel octet), LDS transpose to [B, C, S_i]
template <int ND, typename T>
__global__ __launch_bounds__(256) void hp_col2im_combine_kernel(Geom g, HpDims hd, int S_e,
                                                                const typename SumStore<T>::type *__restrict__ sums,
                                                                typename T::Raw *__restrict__ grad_input) {
  using Raw = typename T::Raw;
  using Sum = typename SumStore<T>::type;
  constexpr bool WIDE = sizeof(Sum) == 4;
  constexpr int L = ND - 1, NS = 1 << L;
  constexpr int QT = 64, CW = 64, TP = QT + 1;
  __shared__ float tile[CW * TP];   // fp32: the stencil sum is rounded once, at the grad_input store
  const int qtiles = (g.S_i + QT - 1) / QT;
  const int b = blockIdx.x / qtiles, q0 = (blockIdx.x - b * qtiles) * QT;
  const int cseg = g.DG == 1 ? hd.Cp : g.Cdg;
  const int W = g.in_sz[L];
  for (int c0 = 0; c0 < hd.Cp; c0 += CW) {
    // 8 lanes per target (64 channels), 32 targets per pass
    for (int it = 0; it < QT / 32; ++it) {
      const int ql = it * 32 + (threadIdx.x >> 3), q = q0 + ql;
      const int c = c0 + (threadIdx.x & 7) * 8;
      float acc[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = 0.f;
      if (q < g.S_i && c < hd.Cp) {
        const int dg = g.DG == 1 ? 0 : c / cseg;
        const int cl = c - dg * cseg;
        int tc[ND], rem = q;
#pragma unroll
        for (int a = L; a > 0; --a) { tc[a] = rem % g.in_sz[a]; rem /= g.in_sz[a]; }
        tc[0] = rem;
        const Sum *base = sums + ((int64_t)(b * g.DG + dg) * S_e * NS) * cseg + cl;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          int er = 0;
#pragma unroll
          for (int ax = 0; ax < L; ++ax) er = er * (g.in_sz[ax] + 1) + tc[ax] + ((s >> (L - 1 - ax)) & 1);
          const Sum *src = base + (((int64_t)er * W + tc[L]) * NS + s) * cseg;
          if constexpr (WIDE) {
            const float4 v0 = reinterpret_cast<const float4 *>(src)[0], v1 = reinterpret_cast<const float4 *>(src)[1];
            acc[0] += v0.x; acc[1] += v0.y; acc[2] += v0.z; acc[3] += v0.w;
            acc[4] += v1.x; acc[5] += v1.y; acc[6] += v1.z; acc[7] += v1.w;
          } else {
            const U4 v = *reinterpret_cast<const U4 *>(src);
            mac8<T>(acc, v, 1.f);
          }
        }
      }
      float *tp = tile + ((threadIdx.x & 7) * 8) * TP + ql;
#pragma unroll
      for (int k = 0; k < 8; ++k) tp[k * TP] = acc[k];
    }
    __syncthreads();
    for (int x = threadIdx.x; x < CW * QT; x += 256) {
      const int cl = x / QT, ql = x - cl * QT;
      const int c = caller_channel(g, c0 + cl), q = q0 + ql;
      if (c >= 0 && q < g.S_i) {
        Raw *dst = grad_input + ((int64_t)b * caller_channels(g) + c) * g.S_i + q;
        const float v = (float)tile[cl * TP + ql];
        T::stf(dst, g.acc_data ? T::ldf(dst) + v : v);
      }
    }
    __syncthreads();
  }
}

template <int ND, typename T>
int launch_col2im(const Geom &g, const HpDims &hd, const Tensors &t, const void *gcol,
                  const int *rowptr, const void *entries, hipStream_t stream) {
  using Raw = typename T::Raw;
  const int cseg = g.DG == 1 ? hd.Cp : g.Cdg;
  const int lanes = (cseg + 7) / 8;
  const int S_e = hp_anchor_space(g);
#define HP_C2I(LPD)                                                                              \
  do {                                                                                           \
    const int qt = 4 * (64 / LPD) * kRun;                                                        \
    hipLaunchKernelGGL((hp_col2im_kernel<ND, T, LPD>), dim3(g.B * ((g.S_i + qt - 1) / qt)),      \
                       dim3(256), 0, stream, g, hd, S_e, (const Raw *)gcol, rowptr,              \
                       (const int4 *)entries, (Raw *)t.grad_input);                              \
  } while (0)
  if (lanes <= 4) HP_C2I(4);
  else if (lanes <= 8) HP_C2I(8);
  else if (lanes <= 16) HP_C2I(16);
  else if (lanes <= 32) HP_C2I(32);
  else HP_C2I(64);
#undef HP_C2I
  return check_launch("hp_col2im");
}

}  // namespace

int hp_csr_zero(const Geom &g, int *cnt, hipStream_t stream) {
  const int64_t cnt_n = (int64_t)g.B * g.DG * hp_anchor_space(g);
  hipLaunchKernelGGL(hp_zero_int_kernel, dim3(grid_for(cnt_n)), dim3(256), 0, stream, cnt, cnt_n);
  return check_launch("hp_zero_cnt");
}

int hp_csr_build(const Geom &g, int dtype, const Tensors &t, int *cnt, int *rowptr, void *entries,
                 hipStream_t stream) {
  const int64_t samples = (int64_t)g.B * g.DG * g.K * g.S_o;
  const int S_e = hp_anchor_space(g);
  int rc;
  hipLaunchKernelGGL(hp_csr_scan_kernel, dim3((S_e + kScanChunk - 1) / kScanChunk, g.B * g.DG), dim3(256), 0, stream,
                     S_e, cnt, rowptr);
  if ((rc = check_launch("hp_csr_scan"))) return rc;
#define HP_CSR(ND, MOD, T)                                                                        \
  hipLaunchKernelGGL((hp_csr_fill_kernel<ND, MOD, T>), dim3(grid_for(samples)), dim3(256), 0, stream, g, S_e, \
                     (const typename T::Raw *)t.offset, (const typename T::Raw *)t.mask, cnt, rowptr,    \
                     (int4 *)entries)
#define HP_CSR_T(T)                                                                               \
  do {                                                                                            \
    if (g.nd == 2) { if (g.modulated) HP_CSR(2, true, T); else HP_CSR(2, false, T); }             \
    else { if (g.modulated) HP_CSR(3, true, T); else HP_CSR(3, false, T); }                       \
  } while (0)
  if (dtype == MDCONV_F16) HP_CSR_T(F16); else HP_CSR_T(BF16);
#undef HP_CSR_T
#undef HP_CSR
  return check_launch("hp_csr_fill");
}

template <int ND, typename T>
static int launch_col2im2(const Geom &g, const HpDims &hd, const Tensors &t, const void *gcol, const int *rowptr,
                   const void *entries, void *sums, hipStream_t stream) {
  using Raw = typename T::Raw;
  const int cseg = g.DG == 1 ? hd.Cp : g.Cdg;
  const int lanes = (cseg + 7) / 8;
  const int S_e = hp_anchor_space(g);
  const int runs_per_seg = (S_e + kRunA - 1) / kRunA;
#define HP_C2S(LPD)                                                                              \
  do {                                                                                           \
    const int runs = 4 * (64 / LPD);                                                             \
    hipLaunchKernelGGL((hp_col2im_sums_kernel<ND, T, LPD>),                                      \
                       dim3(g.B * g.DG * ((runs_per_seg + runs - 1) / runs)), dim3(256), 0, stream, g, hd, S_e, \
                       (const Raw *)gcol, rowptr, (const int4 *)entries, (typename SumStore<T>::type *)sums); \
  } while (0)
  // matrix-core pass 1 for fp16 tensors where a list's channels are 32 / 64 / 128 / 256, the VALU kernel otherwise
  const int nb = cseg / 32;
  bool on_mfma = false;
  if constexpr (ShortEntry<2, T>::value) {   // (F16: the trait names the tensor type)
    if (cseg == 16) {
      const int runs = g.B * g.DG * ((S_e + kRunM - 1) / kRunM);
      const size_t lds = (size_t)4 * (16 * kWtP * 2 + 16 * (cseg + 32) * 2);
      hipLaunchKernelGGL((hp_col2im_sums_mfma_kernel<ND, T, 1, true>), dim3((runs + 3) / 4), dim3(256), lds, stream, g, hd,
                         S_e, (const Raw *)gcol, rowptr, (const int4 *)entries, (typename SumStore<T>::type *)sums);
      on_mfma = true;
    } else if (cseg % 32 == 0 && (nb == 1 || nb == 2 || nb == 4 || nb == 8)) {
      const int runs = g.B * g.DG * ((S_e + kRunM - 1) / kRunM);
      const size_t lds = (size_t)4 * (16 * kWtP * 2 + 16 * (cseg + 32) * 2);
#define HP_C2M(NBV)                                                                              \
      hipLaunchKernelGGL((hp_col2im_sums_mfma_kernel<ND, T, NBV>), dim3((runs + 3) / 4), dim3(256), lds, stream, g, hd, \
                         S_e, (const Raw *)gcol, rowptr, (const int4 *)entries, (typename SumStore<T>::type *)sums)
      if (nb == 1) HP_C2M(1); else if (nb == 2) HP_C2M(2); else if (nb == 4) HP_C2M(4); else HP_C2M(8);
#undef HP_C2M
      on_mfma = true;
    }
  }
  if (on_mfma) {}
  else if (lanes <= 4) HP_C2S(4);
  else if (lanes <= 8) HP_C2S(8);
  else if (lanes <= 16) HP_C2S(16);
  else if (lanes <= 32) HP_C2S(32);
  else HP_C2S(64);
#undef HP_C2S
  int rc = check_launch("hp_col2im_sums");
  if (rc) return rc;
  hipLaunchKernelGGL((hp_col2im_combine_kernel<ND, T>), dim3(g.B * ((g.S_i + 63) / 64)), dim3(256), 0, stream, g, hd,
                     S_e, (const typename SumStore<T>::type *)sums, (Raw *)t.grad_input);
  return check_launch("hp_col2im_combine");
}

size_t hp_col2im_sums_bytes(const Geom &g, const HpDims &hd, int dtype) {
  const size_t cseg = g.DG == 1 ? hd.Cp : g.Cdg;
  return (size_t)g.B * g.DG * hp_anchor_space(g) * (1 << (g.nd - 1)) * cseg * (dtype == MDCONV_BF16 ? 4 : 2);   // SumStore<T>
}

int hp_col2im2(const Geom &g, const HpDims &hd, int dtype, const Tensors &t, const void *gcol,
               const int *rowptr, const void *entries, void *sums, hipStream_t stream) {
  if (dtype == MDCONV_F16)
    return g.nd == 2 ? launch_col2im2<2, F16>(g, hd, t, gcol, rowptr, entries, sums, stream)
                     : launch_col2im2<3, F16>(g, hd, t, gcol, rowptr, entries, sums, stream);
  return g.nd == 2 ? launch_col2im2<2, BF16>(g, hd, t, gcol, rowptr, entries, sums, stream)
                   : launch_col2im2<3, BF16>(g, hd, t, gcol, rowptr, entries, sums, stream);
}

int hp_col2im(const Geom &g, const HpDims &hd, int dtype, const Tensors &t, const void *gcol,
              const int *rowptr, const void *entries, hipStream_t stream) {
  if (dtype == MDCONV_F16)
    return g.nd == 2 ? launch_col2im<2, F16>(g, hd, t, gcol, rowptr, entries, stream)
                     : launch_col2im<3, F16>(g, hd, t, gcol, rowptr, entries, stream);
  return g.nd == 2 ? launch_col2im<2, BF16>(g, hd, t, gcol, rowptr, entries, stream)
                   : launch_col2im<3, BF16>(g, hd, t, gcol, rowptr, entries, stream);
}

}  // namespace mdconv
