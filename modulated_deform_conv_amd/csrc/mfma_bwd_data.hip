// mfma_bwd_data.hip -- grad_offset / grad_mask / grad_input for fp32 on the gfx950 matrix cores.
//
// Reference structure (mdeformable_conv.cu:412-435): GEMM-1 grad_col = W^T . grad_out, then one
// thread per SAMPLE doing 7 global atomics.  Measured on MI355X (tools/ubench_scatter*.hip) the
// 925 M float atomics of cfg2 cost 24.6 ms scattered / 2.8 ms perfectly coalesced, and LDS
// ds_add_f32 is no better (4.7 ms) -- against a 2.3 ms MFMA budget for the whole iteration.
// So nothing here uses floating-point atomics:
//
//  1. mfma_bwd_data_kernel  (col2im_coord + GEMM-1, fused)
//     M = input channels, N = output pixels, K = output channels.  A = W pre-packed in
//     MFMA-fragment order (`pack_wq`), B = grad_out slab through LDS.  In the accumulator layout
//     a lane owns ONE pixel and 32 channels, so grad_offset / grad_mask -- sums over channels of
//     grad_col * d(sample) -- are reduced in registers, then across the two half-waves with one
//     shuffle and across channel-waves through LDS, and written once per (tap, pixel) by their
//     single owner.  grad_col itself is streamed to the workspace CHANNEL-INNERMOST,
//     [b][tap][pix][c] (16-byte stores: a lane holds 4 consecutive channels), for step 3.
//  2. build_scatter_csr  (count -> scan -> fill, integer atomics only)
//     inverts the scatter map: for every (image, input pixel q) the list of
//     (tap * S_o + output pixel, bilinear weight * mask) that land on q.  Depends only on
//     offset / mask.
//  3. col2im_gather_kernel
//     grad_input[b][c][q] += sum over the lists of q of weight * grad_col[b][tap][n][c]:
//     one WAVE per input pixel q, lanes = channels (4 each), so every list entry is one
//     wave-uniform scalar read plus one fully coalesced 16 B/lane vector read of all channels --
//     no divergence, no atomics; a 32-pixel tile is transposed through LDS and added to the
//     NCHW grad_input with whole-line accesses.  (A first version with lanes = pixels and
//     per-lane list walks took 3.5 ms at cfg2; this one is HBM-bound.)
#include "mfma_kernels.hpp"
#include "mfma_tile.hpp"

namespace mdconv {

namespace {

int grid_for(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

// W[o][c][tap] -> wq[tap][ochunk][cblk][q][lane][s] = W[ochunk*16 + 8q + 4(lane>>5) + s]
//                                                      [cblk*32 + (lane&31)][tap]   (0 padded)
__global__ __launch_bounds__(256) void pack_wq_kernel(Geom g, int ochunks, int cblks,
                                                      const float *__restrict__ w,
                                                      float *__restrict__ wq) {
  const int64_t total = (int64_t)g.K * ochunks * cblks * 2 * 64;   // float4 units
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
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
      v[s] = (o < g.O && c < g.C) ? w[((int64_t)o * g.C + c) * g.K + tap] : 0.f;
    }
    reinterpret_cast<float4 *>(wq)[i] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// ---------------------------------------------------------------------------------------------
// 1. GEMM-1 + coordinate gradients + grad_col stream
// ---------------------------------------------------------------------------------------------
// The grad_out tile of the workgroup ([C_out] x BNP pixels, 32 KB at cfg2) does not depend on the
// tap, so it is loaded into LDS ONCE, o-innermost, and the K loops of all taps read their B
// fragments from it with two ds_read_b128 per 16 MFMAs: no global B loads, no LDS writes and no
// barriers inside the tap loop, so the four waves drift apart and one wave's gather epilogue
// overlaps the others' MFMAs.  The per-tap channel reduction goes through a small LDS buffer
// that is flushed every kTapGroup taps (the only barriers left).
constexpr int kTapGroup = 9;

template <int ND, bool MOD, int WAVES_C>
__global__ __launch_bounds__(256, 3) void mfma_bwd_data_kernel(
    Geom g, BwdDims bd, const float *__restrict__ input, const float *__restrict__ gout,
    const float *__restrict__ wq, const float *__restrict__ offset, const float *__restrict__ mask,
    float *__restrict__ gcol, float *__restrict__ grad_offset, float *__restrict__ grad_mask,
    int ntiles) {
  constexpr int NC = 1 << ND, NP = NC / 2;
  constexpr int MB = 2;
  constexpr int WAVES_P = 4 / WAVES_C;
  constexpr int BNP = 32 * WAVES_P;        // pixels per workgroup
  constexpr int RB = 8;                    // accumulator rows gathered per batch
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int T_o = bd.ochunks;              // even
  const int gpitch = T_o * 16 + 4;         // floats per pixel row of the grad_out tile
  float *Gs = smem;                        // [BNP][gpitch]
  float *red = smem + BNP * gpitch;        // [kTapGroup][WAVES_C][ND + 1][BNP]

  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int n0 = tile * BNP;
  const int tid = threadIdx.x, lane = tid & 63, kh = lane >> 5;
  // wave id as an SGPR: anything derived from threadIdx is 'divergent' to hipcc, and a divergent
  // buffer soffset is wrapped in a readfirstlane waterfall per load (cdna_hip_programming.md T20)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave / WAVES_P, wp = wave % WAVES_P;

  // ---- grad_out tile -> LDS (once) ----
  {
    const int j = tid % BNP, osub = tid / BNP;
    constexpr int OSTEP = 256 / BNP;
    const int n_t = min(n0 + j, g.N - 1);
    const int b_t = n_t / g.S_o, pix_t = n_t - b_t * g.S_o;
    const bool t_live = n0 + j < g.N;
    const float *src = gout + ((int64_t)b_t * g.O) * g.S_o + pix_t;
    float *dst = Gs + j * gpitch;
    const int Opad = T_o * 16;
#pragma unroll 8
    for (int o = osub; o < Opad; o += OSTEP) {
      const float v = src[(int64_t)min(o, g.O - 1) * g.S_o];
      dst[o] = (t_live && o < g.O) ? v : 0.f;
    }
  }

  // the pixel this lane owns in the accumulator layout
  const int n_raw = n0 + wp * 32 + (lane & 31);
  const bool live = n_raw < g.N;
  const int n_l = live ? n_raw : g.N - 1;
  const int b_l = n_l / g.S_o, pix_l = n_l - b_l * g.S_o;
  int oc[ND];
  out_coords<ND>(g, pix_l, oc);

  const int passes = bd.cblks_q / (2 * WAVES_C);
  const int frag_bytes = 64 * 16;                      // one [lane][4] fragment
  const int chunk_bytes = bd.cblks_q * 2 * frag_bytes; // one ochunk of wq
  const rsrc_t r_in = make_rsrc(input, (size_t)g.B * g.C * g.S_i * 4);
  const rsrc_t r_wq = make_rsrc(wq, (size_t)g.K * T_o * chunk_bytes);
  const rsrc_t r_gc = make_rsrc(gcol, (size_t)g.B * g.C * g.K * g.S_o * 4);
  const int a_lane = lane * 16;
  const float *Bb = Gs + (wp * 32 + (lane & 31)) * gpitch + 4 * kh;

  auto a_base = [&](int tap, int pass) {
    return tap * T_o * chunk_bytes + ((pass * WAVES_C + wc) * 2) * 2 * frag_bytes;
  };
  auto load_a = [&](float4 (&ra)[MB][2], int soff) {
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#ifdef ABL_BD_NOA
        ra[i][q] = make_float4((float)soff, 1.f, 2.f, (float)a_lane);
#else
        ra[i][q] = buf_load4(r_wq, a_lane + (i * 2 + q) * frag_bytes, soff);
#endif
  };
#ifdef ABL_BD_PRIO_STATIC
  switch ((blockIdx.x >> 8) & 3) { case 0: __builtin_amdgcn_s_setprio(0); break; case 1: __builtin_amdgcn_s_setprio(1); break; case 2: __builtin_amdgcn_s_setprio(2); break; default: __builtin_amdgcn_s_setprio(3); }
#endif
  float4 ra0[MB][2], ra1[MB][2];
  load_a(ra0, a_base(0, 0));
  __syncthreads();   // grad_out tile complete

  for (int tap = 0; tap < g.K; ++tap) {
    // ---- sampling state of (tap, this lane's pixel) ----
    // corner PAIRS (make_pairs): byte offsets, value weights and d/dp weights of both elements
    int voff[NP];
    float w[NC], dw[ND][NC];
    float m = 1.f;
    bool inside;
    {
      float delta[ND];
      const int64_t ob = ((int64_t)b_l * (ND * g.K) + ND * tap) * g.S_o + pix_l;
#pragma unroll
      for (int a = 0; a < ND; ++a) delta[a] = offset[ob + (int64_t)a * g.S_o];
      int tcd[ND];
      tap_coords<ND>(g, tap, tcd);
      TapCoef<ND, float> tc;
      make_tap<ND, float>(g, oc, tcd, delta, true, tc);
      if (MOD) m = mask[((int64_t)b_l * g.K + tap) * g.S_o + pix_l];
      inside = tc.inside;
      int pidx[NP];
      float px[NP], py[NP];
      make_pairs<ND, float>(g, tc, 1.f, pidx, px, py);
#pragma unroll
      for (int pi = 0; pi < NP; ++pi) {
        voff[pi] = (b_l * g.C * g.S_i + pidx[pi] + 4 * kh * g.S_i) * 4;
        w[2 * pi] = px[pi];
        w[2 * pi + 1] = py[pi];
      }
#pragma unroll
      for (int a = 0; a < ND; ++a) {
        make_pairs_d<ND, float>(g, tc, a, px, py);
#pragma unroll
        for (int pi = 0; pi < NP; ++pi) {
          dw[a][2 * pi] = px[pi];
          dw[a][2 * pi + 1] = py[pi];
        }
      }
    }
    const int gc_voff = ((((b_l * g.K + tap) * g.S_o + pix_l) * g.C) + 4 * kh) * 4;
    // S[e] = sum over this lane's channels of grad_col * (element e of the corner pairs).  The corner weights and
    // their derivatives do not depend on the channel, so the epilogue costs 2^ND FMAs per channel
    // and grad_mask / grad_offset are recovered from S once per tap:
    //   grad_mask += sum_ci w[ci] S[ci],   grad_offset_a += m * sum_ci dw[a][ci] S[ci].
    float S[NC];
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) S[ci] = 0.f;

    for (int pass = 0; pass < passes; ++pass) {
      const int cbase = (pass * WAVES_C + wc) * 64;       // this wave's 64 channels
      const int a_soff0 = a_base(tap, pass);
      f32x16 acc[MB];
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

      auto mma = [&](const float4 (&ra)[MB][2], const float *bp) {
        const float4 b0 = *reinterpret_cast<const float4 *>(bp);
        const float4 b1 = *reinterpret_cast<const float4 *>(bp + 8);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const float4 bq = q == 0 ? b0 : b1;
            const float b = s == 0 ? bq.x : (s == 1 ? bq.y : (s == 2 ? bq.z : bq.w));
#pragma unroll
            for (int i = 0; i < MB; ++i) {
              const float a = s == 0 ? ra[i][q].x : (s == 1 ? ra[i][q].y : (s == 2 ? ra[i][q].z : ra[i][q].w));
              acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            }
          }
      };

      // ra0 already holds chunk 0 of this (tap, pass)
#if defined(ABL_BD_PRIO_EPI_HIGH)
      __builtin_amdgcn_s_setprio(0);
#elif defined(ABL_BD_PRIO_EPI_LOW)
      __builtin_amdgcn_s_setprio(3);
#endif
      for (int t = 0; t < T_o; t += 2) {
        load_a(ra1, a_soff0 + (t + 1) * chunk_bytes);
        __builtin_amdgcn_sched_barrier(0);
#ifndef ABL_BD_NOMFMA
        mma(ra0, Bb + t * 16);
#endif
        load_a(ra0, a_soff0 + min(t + 2, T_o - 1) * chunk_bytes);
        __builtin_amdgcn_sched_barrier(0);
        mma(ra1, Bb + (t + 1) * 16);
      }
#if defined(ABL_BD_PRIO_EPI_HIGH)
      __builtin_amdgcn_s_setprio(3);
#elif defined(ABL_BD_PRIO_EPI_LOW)
      __builtin_amdgcn_s_setprio(0);
#endif
      // first A chunk of the next (tap, pass): in flight during the epilogue
      {
        int ntap = tap, npass = pass + 1;
        if (npass == passes) { npass = 0; ntap = min(tap + 1, g.K - 1); }
        load_a(ra0, a_base(ntap, npass));
      }

      // ---- epilogue of (tap, pass): lane = pixel, acc rows = channels ----
      // Rows are processed in batches of RB; the corner gathers of batch k+1 are in flight while
      // batch k is consumed (sched_barriers keep hipcc from hoisting every gather to the top,
      // which cost 284 VGPRs and occupancy 1).
      if (cbase < g.C) {
        constexpr int NBATCH = MB * 16 / RB;
        float2 v[2][RB][NP];
        auto gather_batch = [&](float2 (&vb)[RB][NP], int k) {
          const int mb = (k * RB) / 16, r0 = (k * RB) % 16;
#pragma unroll
          for (int rr = 0; rr < RB; ++rr) {
            const int r = r0 + rr;
            const int cu = cbase + mb * 32 + (r & 3) + 8 * (r >> 2);   // + 4*kh is in the voffset
            const int cs = min(cu, g.C - 5) * g.S_i * 4;   // cu % 8 < 4, so C-5 is the last valid one
#pragma unroll
#ifdef ABL_BD_NOGATHER
            for (int pi = 0; pi < NP; ++pi) vb[rr][pi] = make_float2(w[0], (float)cs);
#else
            for (int pi = 0; pi < NP; ++pi) vb[rr][pi] = buf_load2(r_in, voff[pi], cs);
#endif
          }
        };
        gather_batch(v[0], 0);
#pragma unroll
        for (int k = 0; k < NBATCH; ++k) {
          if (k + 1 < NBATCH) gather_batch(v[(k + 1) & 1], k + 1);
          asm volatile("" ::: "memory");   // IR-level fence: later gathers must not be hoisted here
          __builtin_amdgcn_sched_barrier(0);
          const int mb = (k * RB) / 16, r0 = (k * RB) % 16;
          // grad_col[b][tap][pix][c]: rows r0+4g .. r0+4g+3 are 4 consecutive channels.  No
          // branches here (they would split the block and let hipcc sink the S updates below all
          // four batches, keeping 128 gathered values live): dead lanes / padded channels store to
          // an out-of-range offset, which the buffer bounds check drops.
#pragma unroll
          for (int gq = 0; gq < RB / 4; ++gq) {
            const int cu4 = cbase + mb * 32 + 8 * ((r0 >> 2) + gq);
            const int vo = (live && cu4 < g.C) ? gc_voff : (int)0x7ffffff0;
#ifdef ABL_BD_NOSTORE
            if (acc[mb][r0 + 4 * gq] == 123.456f)
#endif
#ifdef ABL_BD_NT
            buf_store4<2>
#else
            buf_store4
#endif
                      (r_gc, vo, cu4 * 4, acc[mb][r0 + 4 * gq], acc[mb][r0 + 4 * gq + 1],
                       acc[mb][r0 + 4 * gq + 2], acc[mb][r0 + 4 * gq + 3]);
          }
          // padded channels have grad_col == 0 exactly (zero weight rows), no predicate needed
#pragma unroll
          for (int rr = 0; rr < RB; ++rr) {
            const float gc = acc[mb][r0 + rr];
#pragma unroll
            for (int pi = 0; pi < NP; ++pi) {
              S[2 * pi] = fmaf(gc, v[k & 1][rr][pi].x, S[2 * pi]);
              S[2 * pi + 1] = fmaf(gc, v[k & 1][rr][pi].y, S[2 * pi + 1]);
            }
          }
#pragma unroll
          for (int ci = 0; ci < NC; ++ci) asm volatile("" : "+v"(S[ci]));   // pin the updates here
          asm volatile("" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }

    float goff[ND], gm = 0.f;
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) gm = fmaf(w[ci], S[ci], gm);
    const float mg = (!g.range_gate || inside) ? m : 0.f;
#pragma unroll
    for (int a = 0; a < ND; ++a) {
      goff[a] = 0.f;
#pragma unroll
      for (int ci = 0; ci < NC; ++ci) goff[a] = fmaf(dw[a][ci], S[ci], goff[a]);
      goff[a] *= mg;
    }
    // ---- reduce over channels: the two half-waves here, the WAVES_C channel-waves at the flush ----
#pragma unroll
    for (int a = 0; a < ND; ++a) goff[a] += __shfl_xor(goff[a], 32, 64);
    gm += __shfl_xor(gm, 32, 64);
    const int slot = tap % kTapGroup;
    if (kh == 0) {
      float *rp = red + ((slot * WAVES_C + wc) * (ND + 1)) * BNP + wp * 32 + lane;
#pragma unroll
      for (int a = 0; a < ND; ++a) rp[a * BNP] = goff[a];
      rp[ND * BNP] = gm;
    }
    if (slot == kTapGroup - 1 || tap == g.K - 1) {
      __syncthreads();
      // single owner of every (b, tap, pix): plain accumulate (the C ABI accumulates into grads)
      const int tap0 = tap - slot;
      const int items = (slot + 1) * (ND + 1) * BNP;
      for (int it = tid; it < items; it += 256) {
        const int jj = it % BNP, a = (it / BNP) % (ND + 1), sl = it / (BNP * (ND + 1));
        const int n = n0 + jj;
        if (n < g.N && (MOD || a < ND)) {
          float sum = 0.f;
#pragma unroll
          for (int x = 0; x < WAVES_C; ++x) sum += red[((sl * WAVES_C + x) * (ND + 1) + a) * BNP + jj];
          const int b = n / g.S_o, pix = n - b * g.S_o, tp = tap0 + sl;
          if (a < ND) grad_offset[((int64_t)b * (ND * g.K) + ND * tp + a) * g.S_o + pix] += sum;
          else grad_mask[((int64_t)b * g.K + tp) * g.S_o + pix] += sum;
        }
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------
// 2. inverse scatter map (CSR keyed by (b, tap, q)); DG == 1
// ---------------------------------------------------------------------------------------------
template <int ND, bool MOD, bool FILL>
__global__ __launch_bounds__(256) void csr_pass_kernel(Geom g, const float *__restrict__ offset,
                                                       const float *__restrict__ mask,
                                                       int *__restrict__ cnt_or_cursor,
                                                       const int *__restrict__ rowptr,
                                                       int2 *__restrict__ entries) {
  constexpr int NC = 1 << ND;
  const int64_t total = (int64_t)g.B * g.K * g.S_o;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int pix = (int)(i % g.S_o);
    const int tap = (int)((i / g.S_o) % g.K);
    const int b = (int)(i / g.S_o / g.K);
    int oc[ND], tcd[ND];
    out_coords<ND>(g, pix, oc);
    tap_coords<ND>(g, tap, tcd);
    float delta[ND];
    const int64_t ob = ((int64_t)b * (ND * g.K) + ND * tap) * g.S_o + pix;
#pragma unroll
    for (int a = 0; a < ND; ++a) delta[a] = offset[ob + (int64_t)a * g.S_o];
    TapCoef<ND, float> tc;
    make_tap<ND, float>(g, oc, tcd, delta, true, tc);
    const float m = MOD ? mask[((int64_t)b * g.K + tap) * g.S_o + pix] : 1.f;
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) {
      const float wa = corner_weight_atom<ND, float>(tc, ci);   // validity folded in
      if (wa != 0.f) {
        const int q = corner_index<ND, float>(tc, ci);
        if (!FILL) {
          atomicAdd(cnt_or_cursor + (int64_t)b * g.S_i + q, 1);
        } else {
          const int pos = rowptr[(int64_t)b * (g.S_i + 1) + q] + atomicAdd(cnt_or_cursor + (int64_t)b * g.S_i + q, 1);
          entries[(int64_t)b * ((int64_t)g.K * g.S_o * NC) + pos] =
              make_int2(tap * g.S_o + pix, __float_as_int(wa * m));
        }
      }
    }
  }
}

// exclusive scan of cnt[seg][0..S_i) -> rowptr[seg][0..S_i], one workgroup per segment
__global__ __launch_bounds__(256) void csr_scan_kernel(int S_i, const int *__restrict__ cnt,
                                                       int *__restrict__ rowptr) {
  __shared__ int wsum[4];
  __shared__ int carry;
  const int seg = blockIdx.x;
  const int *c = cnt + (int64_t)seg * S_i;
  int *rp = rowptr + (int64_t)seg * (S_i + 1);
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < S_i; base += 256) {
    const int i = base + threadIdx.x;
    const int v = i < S_i ? c[i] : 0;
    int x = v;   // inclusive scan inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int y = __shfl_up(x, d, 64);
      if ((threadIdx.x & 63) >= d) x += y;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
    __syncthreads();
    int woff = 0;
    for (int k = 0; k < (int)(threadIdx.x >> 6); ++k) woff += wsum[k];
    const int excl = carry + woff + x - v;
    if (i < S_i) rp[i] = excl;
    __syncthreads();
    if (threadIdx.x == 255) carry = excl + v;
    __syncthreads();
  }
  if (threadIdx.x == 0) rp[S_i] = carry;
}

// ---------------------------------------------------------------------------------------------
// 3. grad_input[b][c][q] += sum_{e in list(b, q)} w_e * gcol[b][src_e][c],  src = tap * S_o + pix
// workgroup = 32 consecutive q of one image x 256 channels; wave w walks q = q0 + w, w + 4, ...
// The list of q is fetched by ONE coalesced vector load (lane i <- entry i) and broadcast with
// readlane, so there is no dependent scalar-load chain per entry.
// ---------------------------------------------------------------------------------------------
template <int ND>
__global__ __launch_bounds__(256) void col2im_gather_kernel(Geom g, const float *__restrict__ gcol,
                                                            const int *__restrict__ rowptr,
                                                            const int2 *__restrict__ entries,
                                                            float *__restrict__ grad_input) {
  constexpr int NC = 1 << ND;
  constexpr int QT = 32;
  __shared__ float tile[256 * (QT + 1)];   // [c][q], pitch 33
  const int qtiles = (g.S_i + QT - 1) / QT;
  // every grad_col row is read by up to 2^ND targets (q, q+1, q+W, ...): keep neighbouring q
  // tiles on ONE XCD so those re-reads hit its L2 instead of HBM (3.7 GB -> ~1 GB at cfg2)
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int b = bid / qtiles;
  const int q0 = (bid - b * qtiles) * QT;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const rsrc_t r_gc = make_rsrc(gcol + (size_t)b * g.K * g.S_o * g.C, (size_t)g.K * g.S_o * g.C * 4);
  const int *rp = rowptr + (int64_t)b * (g.S_i + 1);
  const int2 *ent = entries + (int64_t)b * ((int64_t)g.K * g.S_o * NC);
  for (int cb = blockIdx.y * 256; cb < g.C; cb += gridDim.y * 256) {
    const int c4 = cb + lane * 4;
    const int c_voff = (c4 < g.C ? c4 : 0) * 4;   // C % 4 == 0
    for (int qi = wave; qi < QT; qi += 4) {
      const int q = q0 + qi;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < g.S_i) {
        const int e0 = __builtin_amdgcn_readfirstlane(rp[q]);
        const int e1 = __builtin_amdgcn_readfirstlane(rp[q + 1]);
        for (int base = e0; base < e1; base += 64) {
          const int cnt = min(64, e1 - base);
          const int2 mine = (lane < cnt) ? ent[base + lane] : make_int2(0, 0);
          // 16 independent row loads in flight per step (a one-entry-at-a-time loop serialises
          // the full memory latency per entry: 0.65 ms at cfg2); lanes >= cnt hold weight 0, row 0
          for (int i = 0; i < cnt; i += 16) {
            float4 v[16];
            float we[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
              const int src = __builtin_amdgcn_readlane(mine.x, (i + u) & 63);
              we[u] = __int_as_float(__builtin_amdgcn_readlane(mine.y, (i + u) & 63));
              v[u] = buf_load4(r_gc, c_voff, src * g.C * 4);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
              acc.x = fmaf(we[u], v[u].x, acc.x); acc.y = fmaf(we[u], v[u].y, acc.y);
              acc.z = fmaf(we[u], v[u].z, acc.z); acc.w = fmaf(we[u], v[u].w, acc.w);
            }
          }
        }
      }
      float *tp = tile + (lane * 4) * (QT + 1) + qi;
      tp[0] = acc.x; tp[QT + 1] = acc.y; tp[2 * (QT + 1)] = acc.z; tp[3 * (QT + 1)] = acc.w;
    }
    __syncthreads();
    // transpose out: thread = (channel, 8 consecutive q); a wave covers 8 channels x 32 q
    {
      const int cl = threadIdx.x >> 2, qs = (threadIdx.x & 3) * 8;
      for (int cc = cl; cc < 256; cc += 64) {
        const int c = cb + cc;
        if (c < g.C) {
          float *dst = grad_input + ((int64_t)b * g.C + c) * g.S_i + q0 + qs;
          const float *src = tile + cc * (QT + 1) + qs;
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if (q0 + qs + k < g.S_i) dst[k] += src[k];
        }
      }
    }
    __syncthreads();
  }
}

}  // namespace

int pack_wq_f32(const Geom &g, const BwdDims &bd, const float *weight, float *wq, hipStream_t stream) {
  const int64_t total = (int64_t)g.K * bd.ochunks * bd.cblks_q * 2 * 64;
  hipLaunchKernelGGL(pack_wq_kernel, dim3(grid_for(total)), dim3(256), 0, stream, g, bd.ochunks,
                     bd.cblks_q, weight, wq);
  return check_launch("pack_wq");
}

size_t bwd_data_lds_bytes(const Geom &g, const BwdDims &bd) {
  const int bnp = 32 * (4 / bd.waves_c);
  return ((size_t)bnp * (bd.ochunks * 16 + 4) + (size_t)kTapGroup * 128 * (g.nd + 1)) * sizeof(float);
}

int mfma_bwd_data_f32(const Geom &g, const BwdDims &bd, const Tensors &t, const float *wq,
                      float *gcol, hipStream_t stream) {
#define LAUNCH_BD(ND, MOD, WC)                                                                  \
  do {                                                                                          \
    const int bnp = 32 * (4 / WC);                                                              \
    const int ntiles = (g.N + bnp - 1) / bnp;                                                   \
    const size_t lds = bwd_data_lds_bytes(g, bd);                                               \
    if (lds > 64 * 1024) {                                                                      \
      hipError_t ea = hipFuncSetAttribute((const void *)mfma_bwd_data_kernel<ND, MOD, WC>,      \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (ea != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(ea)); return MDCONV_ELAUNCH; } \
    }                                                                                           \
    hipLaunchKernelGGL((mfma_bwd_data_kernel<ND, MOD, WC>), dim3(ntiles), dim3(256), lds, stream, \
                       g, bd, (const float *)t.input, (const float *)t.grad_output, wq,         \
                       (const float *)t.offset, (const float *)t.mask, gcol,                    \
                       (float *)t.grad_offset, (float *)t.grad_mask, ntiles);                   \
  } while (0)
#define LAUNCH_BD2(ND, MOD)                                                                     \
  do {                                                                                          \
    if (bd.waves_c == 4) LAUNCH_BD(ND, MOD, 4);                                                 \
    else if (bd.waves_c == 2) LAUNCH_BD(ND, MOD, 2);                                            \
    else LAUNCH_BD(ND, MOD, 1);                                                                 \
  } while (0)
  if (g.nd == 2) { if (g.modulated) LAUNCH_BD2(2, true); else LAUNCH_BD2(2, false); }
  else { if (g.modulated) LAUNCH_BD2(3, true); else LAUNCH_BD2(3, false); }
#undef LAUNCH_BD2
#undef LAUNCH_BD
  return check_launch("mfma_bwd_data");
}

int col2im_f32(const Geom &g, const BwdDims &bd, const Tensors &t, const float *gcol, int *cnt,
               int *rowptr, void *entries, hipStream_t stream) {
  const int64_t samples = (int64_t)g.B * g.K * g.S_o;
  const size_t cnt_bytes = (size_t)g.B * g.S_i * sizeof(int);
  hipError_t e = hipMemsetAsync(cnt, 0, cnt_bytes, stream);
  if (e != hipSuccess) { set_error("hipMemsetAsync failed: %s", hipGetErrorString(e)); return MDCONV_ELAUNCH; }
#define LAUNCH_CSR(ND, MOD, FILL)                                                               \
  hipLaunchKernelGGL((csr_pass_kernel<ND, MOD, FILL>), dim3(grid_for(samples)), dim3(256), 0,   \
                     stream, g, (const float *)t.offset, (const float *)t.mask, cnt, rowptr,    \
                     (int2 *)entries)
#define LAUNCH_CSR2(FILL)                                                                       \
  do {                                                                                          \
    if (g.nd == 2) { if (g.modulated) LAUNCH_CSR(2, true, FILL); else LAUNCH_CSR(2, false, FILL); } \
    else { if (g.modulated) LAUNCH_CSR(3, true, FILL); else LAUNCH_CSR(3, false, FILL); }      \
  } while (0)
  LAUNCH_CSR2(false);
  int rc = check_launch("csr_count");
  if (rc) return rc;
  hipLaunchKernelGGL(csr_scan_kernel, dim3(g.B), dim3(256), 0, stream, g.S_i, cnt, rowptr);
  if ((rc = check_launch("csr_scan"))) return rc;
  e = hipMemsetAsync(cnt, 0, cnt_bytes, stream);
  if (e != hipSuccess) { set_error("hipMemsetAsync failed: %s", hipGetErrorString(e)); return MDCONV_ELAUNCH; }
  LAUNCH_CSR2(true);
  if ((rc = check_launch("csr_fill"))) return rc;
#undef LAUNCH_CSR2
#undef LAUNCH_CSR
  const int qtiles = (g.S_i + 31) / 32;
  const dim3 grid(g.B * qtiles, 1);
  if (g.nd == 2)
    hipLaunchKernelGGL((col2im_gather_kernel<2>), grid, dim3(256), 0, stream, g, gcol, rowptr,
                       (const int2 *)entries, (float *)t.grad_input);
  else
    hipLaunchKernelGGL((col2im_gather_kernel<3>), grid, dim3(256), 0, stream, g, gcol, rowptr,
                       (const int2 *)entries, (float *)t.grad_input);
  (void)bd;
  return check_launch("col2im_gather");
}

}  // namespace mdconv
