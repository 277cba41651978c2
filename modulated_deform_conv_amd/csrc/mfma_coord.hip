// mfma_coord.hip -- the "split drain" of the fp32 backward (round 4).
//
// Reference: the per-sample gradient kernel (mdeformable_conv.cu:202-318; deformable_conv.cu:198-287;
// deformable_conv3d.cu:259-389; mdeformable_conv3d.cu:265-395) does three things with the grad_col value of a
// sample: scatter it to grad_input, and add its products with the corner values to grad_offset / grad_mask
// (C / DG same-address atomics per (tap, pixel)).  Rounds 1-3 ran the latter INSIDE GEMM-1 (mfma_bwd_data.hip):
// the accumulators of a tap were drained against the gathered corners while the next tap's K loop ran.  That
// kept the kernel at 253 registers, two waves per SIMD, and 0.69 of the matrix peak at cfg2 -- the same loop
// with the drain removed measures 0.88-0.92 ms against 1.08.  Here the drain is its own memory-bound kernel that
// runs BESIDE GEMM-2 on the forked stream, where the grad_input gather already lives:
//
//   tap_prepass_kernel   one thread per (image, deformable group, tap, pixel): the tap table of the
//                        channels-last GEMM-2 (2^ND corner byte offsets into xt + 2^ND weights, mask folded in)
//                        and the counting pass of the inverted scatter map.  Needs offset / mask only, so it
//                        runs beside GEMM-1.
//   coord_grad_kernel    grad_offset / grad_mask of a (tap, pixel) = sums over the channels of its deformable
//                        group of  grad_col[c] * corner[ci][c],  combined with the corner weights:
//                          grad_mask  = sum_ci w[ci] S[ci]
//                          grad_off_a = mask * sum_ci dw_a[ci] S[ci]       (only inside the image, quirk Q2)
//                        A wave owns 64 consecutive pixels of one (deformable group, tap):
//                          A  lane = pixel: sampling state; corner byte offsets and the grad_col row offset go
//                             to a wave-private LDS table, weights stay in the lane's registers;
//                          B  16 lanes per pixel, four pixels at a time: 16-byte pieces of the grad_col row and of
//                             the 2^ND corner rows of xt (every aligned quad of lanes reads 64 contiguous bytes --
//                             the fast case of the texture path), 4 * 2^ND FMAs per piece, DPP reduction over the
//                             16 lanes, S[ci] back to LDS;
//                          C  lane = pixel again: the two combinations above, one coalesced store per tensor row.
//                        No atomics: every (image, group, tap, pixel) has one owner.
//   Invalid corners (outside the image, or gated by `d > EPS` in the files that gate their loads) are parked out of
//   the buffer's range, so they are never read -- like the reference, and unlike the pair loads of the fused drain.
#include "mfma_kernels.hpp"
#include "mfma_tile.hpp"

namespace mdconv {

namespace {

constexpr int kOobC = 0x7ffffff0;   // out-of-range buffer offset: loads give 0

int grid_for_c(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

template <int ND, bool MOD>
__global__ __launch_bounds__(256) void tap_prepass_kernel(Geom g, int Np, int S_e, int sample_keyed,
                                                          const float *__restrict__ offset,
                                                          const float *__restrict__ mask,
                                                          int *__restrict__ cnt, int *__restrict__ table) {
  constexpr int NC = 1 << ND, NP = NC / 2;
  // index = ((dg * K + tap) * Np + n): the table's own order, so the 2 * NC-word entries of a wave are contiguous
  const int64_t total = (int64_t)g.DG * g.K * Np;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int n = (int)(i % Np);
    const int tap = (int)((i / Np) % g.K);
    const int dg = (int)(i / Np / g.K);
    int ev[2 * NC];
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) { ev[ci] = kOobC; ev[NC + ci] = 0; }
    if (n < g.N) {
      const int b = n / g.S_o, pix = n - b * g.S_o;
      const int seg = b * g.DG + dg;
      int oc[ND], tcd[ND];
      out_coords<ND>(g, pix, oc);
      tap_coords<ND>(g, tap, tcd);
      float delta[ND];
      const int64_t ob = ((int64_t)seg * (ND * g.K) + ND * tap) * g.S_o + pix;
#pragma unroll
      for (int a = 0; a < ND; ++a) delta[a] = offset[ob + (int64_t)a * g.S_o];
      const float m = MOD ? mask[((int64_t)seg * g.K + tap) * g.S_o + pix] : 1.f;
      TapCoef<ND, float> tc;
      make_tap<ND, float>(g, oc, tcd, delta, true, tc);
#pragma unroll
      for (int ci = 0; ci < NC; ++ci)
        if (corner_is_read<ND, float>(tc, ci)) {
          ev[ci] = (b * g.S_i + corner_index<ND, float>(tc, ci)) * g.C * 4;
          ev[NC + ci] = __float_as_int(corner_weight<ND, float>(tc, ci) * m);
        }
      // counting pass of the inverted scatter map: exactly what the fill passes will insert
      // (csr_fill_kernel: corner pairs keyed by their first element; csr_fill3d_kernel: samples keyed by their
      // low corner in the extended anchor space)
      if (sample_keyed) {
        if constexpr (ND == 3) {
          SampleAnchor<ND> sa;
          sample_anchor<ND>(g, tc, 1.f, sa);
          if (sa.on) atomicAdd(cnt + (int64_t)seg * S_e + sa.qa, 1);
        }
      } else {
        int aidx[NP];
        float ax[NP], ay[NP];
        make_pairs_f<ND, float>(g, tc, tc.wl, tc.wha, 1.f, aidx, ax, ay);
        int *cseg = cnt + (int64_t)seg * g.S_i;
#pragma unroll
        for (int pi = 0; pi < NP; ++pi)
          if (ax[pi] != 0.f || ay[pi] != 0.f) atomicAdd(cseg + aidx[pi], 1);
      }
    }
    int4 *e = reinterpret_cast<int4 *>(table + i * (2 * NC));
#pragma unroll
    for (int q4 = 0; q4 < 2 * NC; q4 += 4) e[q4 / 4] = make_int4(ev[q4], ev[q4 + 1], ev[q4 + 2], ev[q4 + 3]);
  }
}

// sum over the 16 lanes of a DPP row; every lane of the row ends up with the total
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
  return v;
}

template <int ND, bool MOD>
__global__ __launch_bounds__(256) void coord_grad_kernel(Geom g, const float *__restrict__ xt,
                                                         const float *__restrict__ gcol,
                                                         const float *__restrict__ offset,
                                                         const float *__restrict__ mask,
                                                         float *__restrict__ grad_offset,
                                                         float *__restrict__ grad_mask, int nblocks64) {
  constexpr int NC = 1 << ND;
  constexpr int ROW = NC + 1;                       // LDS words per pixel: 2^ND corner offsets + the grad_col row
  __shared__ int Soff[4][64 * ROW];
  __shared__ float Ssum[4][64 * NC];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int *soff = Soff[wave];
  float *ssum = Ssum[wave];
  const rsrc_t r_xt = make_rsrc(xt, (size_t)g.B * g.S_i * g.C * sizeof(float));
  const rsrc_t r_gc = make_rsrc(gcol, (size_t)g.B * g.K * g.S_o * g.C * sizeof(float));
  const int steps = (g.DG == 1 ? g.C : g.Cdg) / 64;   // 64-channel pieces of a deformable group (C, C_dg % 64 == 0)
  const int sub = lane >> 4, l16 = lane & 15;
  // unit = ((block of 64 pixels) * K + tap) * DG + dg: the taps and groups of one pixel block are neighbours, so the
  // corner rows they share stay in the XCD's L2
  const int64_t units = (int64_t)nblocks64 * g.K * g.DG;
  for (int64_t u = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * 4 + wave; u < units; u += (int64_t)gridDim.x * 4) {
    const int dg = (int)(u % g.DG);
    const int tap = (int)((u / g.DG) % g.K);
    const int nb = (int)(u / g.DG / g.K);
    // ---- A: lane = pixel ----
    const int n = nb * 64 + lane;
    const bool live = n < g.N;
    const int n_l = live ? n : g.N - 1;
    const int b = n_l / g.S_o, pix = n_l - b * g.S_o;
    const int seg = b * g.DG + dg;
    int oc[ND], tcd[ND];
    out_coords<ND>(g, pix, oc);
    tap_coords<ND>(g, tap, tcd);
    float delta[ND];
    const int64_t ob = ((int64_t)seg * (ND * g.K) + ND * tap) * g.S_o + pix;
#pragma unroll
    for (int a = 0; a < ND; ++a) delta[a] = offset[ob + (int64_t)a * g.S_o];
    const int64_t mb = ((int64_t)seg * g.K + tap) * g.S_o + pix;
    const float m = MOD ? mask[mb] : 1.f;
    TapCoef<ND, float> tc;
    make_tap<ND, float>(g, oc, tcd, delta, true, tc);
    const float mg = (!g.range_gate || tc.inside) ? m : 0.f;
    float w[NC], dw[ND][NC];
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) {
      const bool rd = live && corner_is_read<ND, float>(tc, ci);
      soff[lane * ROW + ci] = rd ? ((b * g.S_i + corner_index<ND, float>(tc, ci)) * g.C + dg * (steps * 64)) * 4 : kOobC;
      w[ci] = corner_weight<ND, float>(tc, ci);
#pragma unroll
      for (int a = 0; a < ND; ++a) dw[a][ci] = corner_dweight<ND, float>(tc, ci, a);
    }
    soff[lane * ROW + NC] = live ? (((b * g.K + tap) * g.S_o + pix) * g.C + dg * (steps * 64)) * 4 : kOobC;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private LDS: program order is enough
    // ---- B: 16 lanes per pixel, four pixels per iteration ----
    for (int it = 0; it < 16; ++it) {
      const int p = it * 4 + sub;
      int off[ROW];
#pragma unroll
      for (int k = 0; k < ROW; ++k) off[k] = soff[p * ROW + k];
      float S[NC];
#pragma unroll
      for (int ci = 0; ci < NC; ++ci) S[ci] = 0.f;
      for (int s0 = 0; s0 < steps; s0 += 2) {
        const bool two = s0 + 1 < steps;
        const int v0 = (s0 * 64 + l16 * 4) * 4, v1 = two ? v0 + 256 : kOobC;
        const float4 g0 = buf_load4(r_gc, off[NC] + v0, 0);
        const float4 g1 = buf_load4(r_gc, off[NC] + v1, 0);
        float4 x0[NC], x1[NC];
#pragma unroll
        for (int ci = 0; ci < NC; ++ci) {
          x0[ci] = buf_load4(r_xt, off[ci] + v0, 0);
          x1[ci] = buf_load4(r_xt, off[ci] + v1, 0);
        }
#pragma unroll
        for (int ci = 0; ci < NC; ++ci) {
          S[ci] = fmaf(g0.w, x0[ci].w, fmaf(g0.z, x0[ci].z, fmaf(g0.y, x0[ci].y, fmaf(g0.x, x0[ci].x, S[ci]))));
          S[ci] = fmaf(g1.w, x1[ci].w, fmaf(g1.z, x1[ci].z, fmaf(g1.y, x1[ci].y, fmaf(g1.x, x1[ci].x, S[ci]))));
        }
      }
#pragma unroll
      for (int ci = 0; ci < NC; ++ci) S[ci] = row16_sum(S[ci]);
      if (l16 == 0) {
#pragma unroll
        for (int ci = 0; ci < NC; ++ci) ssum[p * NC + ci] = S[ci];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---- C: lane = pixel ----
    if (live) {
      float gm = 0.f, goff[ND];
#pragma unroll
      for (int a = 0; a < ND; ++a) goff[a] = 0.f;
#pragma unroll
      for (int ci = 0; ci < NC; ++ci) {
        const float s = ssum[lane * NC + ci];
        gm = fmaf(w[ci], s, gm);
#pragma unroll
        for (int a = 0; a < ND; ++a) goff[a] = fmaf(dw[a][ci], s, goff[a]);
      }
#pragma unroll
      for (int a = 0; a < ND; ++a) {
        float *dst = grad_offset + ob + (int64_t)a * g.S_o;
        const float v = goff[a] * mg;
        *dst = g.acc_data ? *dst + v : v;
      }
      if (MOD) {
        float *dst = grad_mask + mb;
        *dst = g.acc_data ? *dst + gm : gm;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the next unit overwrites the wave's tables
  }
}

// split_drain == 2: the corner sums come from the channels-last GEMM-2 (mfma_bwd_weight_cl.hip, COORD) as one
// partial per 64-channel block, sbuf[cblk][tap][n][ci]; one thread per (image, deformable group, tap, pixel) adds the
// blocks of its group and applies the weights.  Same corner order as tap_prepass_kernel's table (corner ci).
template <int ND, bool MOD>
__global__ __launch_bounds__(256) void coord_finish_kernel(Geom g, int Np, int bpd, const float *__restrict__ sbuf,
                                                           const float *__restrict__ offset,
                                                           const float *__restrict__ mask,
                                                           float *__restrict__ grad_offset,
                                                           float *__restrict__ grad_mask) {
  constexpr int NC = 1 << ND;
  const int64_t total = (int64_t)g.DG * g.K * g.N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int n = (int)(i % g.N);
    const int tap = (int)((i / g.N) % g.K);
    const int dg = (int)(i / g.N / g.K);
    const int b = n / g.S_o, pix = n - b * g.S_o;
    const int seg = b * g.DG + dg;
    int oc[ND], tcd[ND];
    out_coords<ND>(g, pix, oc);
    tap_coords<ND>(g, tap, tcd);
    float delta[ND];
    const int64_t ob = ((int64_t)seg * (ND * g.K) + ND * tap) * g.S_o + pix;
#pragma unroll
    for (int a = 0; a < ND; ++a) delta[a] = offset[ob + (int64_t)a * g.S_o];
    const int64_t mb = ((int64_t)seg * g.K + tap) * g.S_o + pix;
    const float m = MOD ? mask[mb] : 1.f;
    TapCoef<ND, float> tc;
    make_tap<ND, float>(g, oc, tcd, delta, true, tc);
    const float mg = (!g.range_gate || tc.inside) ? m : 0.f;
    float S[NC];
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) S[ci] = 0.f;
    for (int k = 0; k < bpd; ++k) {
      const float4 *src = reinterpret_cast<const float4 *>(sbuf + (((size_t)(dg * bpd + k) * g.K + tap) * Np + n) * NC);
#pragma unroll
      for (int c4 = 0; c4 < NC; c4 += 4) {
        const float4 v = src[c4 / 4];
        S[c4] += v.x; S[c4 + 1] += v.y; S[c4 + 2] += v.z; S[c4 + 3] += v.w;
      }
    }
    float gm = 0.f;
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) gm = fmaf(corner_weight<ND, float>(tc, ci), S[ci], gm);
#pragma unroll
    for (int a = 0; a < ND; ++a) {
      float go = 0.f;
#pragma unroll
      for (int ci = 0; ci < NC; ++ci) go = fmaf(corner_dweight<ND, float>(tc, ci, a), S[ci], go);
      float *dst = grad_offset + ob + (int64_t)a * g.S_o;
      const float v = go * mg;
      *dst = g.acc_data ? *dst + v : v;
    }
    if (MOD) {
      float *dst = grad_mask + mb;
      *dst = g.acc_data ? *dst + gm : gm;
    }
  }
}

}  // namespace

int coord_finish_f32(const Geom &g, const BwdDims &bd, const Tensors &t, const float *sbuf, hipStream_t stream) {
  const int64_t total = (int64_t)g.DG * g.K * g.N;
  const int bpd = g.DG == 1 ? bd.cblks : g.Cdg / 64;   // 64-channel blocks per deformable group
#define LAUNCH_CF(ND, MOD)                                                                                    \
  hipLaunchKernelGGL((coord_finish_kernel<ND, MOD>), dim3(grid_for_c(total)), dim3(256), 0, stream, g, bd.Np,  \
                     bpd, sbuf, (const float *)t.offset, (const float *)t.mask, (float *)t.grad_offset,       \
                     (float *)t.grad_mask)
  if (g.nd == 2) { if (g.modulated) LAUNCH_CF(2, true); else LAUNCH_CF(2, false); }
  else { if (g.modulated) LAUNCH_CF(3, true); else LAUNCH_CF(3, false); }
#undef LAUNCH_CF
  return check_launch("coord_finish");
}

int tap_prepass_f32(const Geom &g, const BwdDims &bd, const Tensors &t, int *cnt, int *table, hipStream_t stream) {
  const int64_t total = (int64_t)g.DG * g.K * bd.Np;
#define LAUNCH_TP(ND, MOD)                                                                                    \
  hipLaunchKernelGGL((tap_prepass_kernel<ND, MOD>), dim3(grid_for_c(total)), dim3(256), 0, stream, g, bd.Np,  \
                     bd.S_e, bd.sample_keyed, (const float *)t.offset, (const float *)t.mask, cnt, table)
  if (g.nd == 2) { if (g.modulated) LAUNCH_TP(2, true); else LAUNCH_TP(2, false); }
  else { if (g.modulated) LAUNCH_TP(3, true); else LAUNCH_TP(3, false); }
#undef LAUNCH_TP
  return check_launch("tap_prepass");
}

int coord_grad_f32(const Geom &g, const BwdDims &bd, const Tensors &t, const float *gcol, const float *xt,
                   hipStream_t stream) {
  const int nblocks64 = (g.N + 63) / 64;
  const int64_t units = (int64_t)nblocks64 * g.K * g.DG;
  int64_t blocks = (units + 3) / 4;
  const int64_t cap = (int64_t)device_cus() * 16;   // a few resident workgroups per CU walk the unit list
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
#define LAUNCH_CG(ND, MOD)                                                                                    \
  hipLaunchKernelGGL((coord_grad_kernel<ND, MOD>), dim3((unsigned)blocks), dim3(256), 0, stream, g, xt, gcol, \
                     (const float *)t.offset, (const float *)t.mask, (float *)t.grad_offset,                  \
                     (float *)t.grad_mask, nblocks64)
  if (g.nd == 2) { if (g.modulated) LAUNCH_CG(2, true); else LAUNCH_CG(2, false); }
  else { if (g.modulated) LAUNCH_CG(3, true); else LAUNCH_CG(3, false); }
#undef LAUNCH_CG
  return check_launch("coord_grad");
}

}  // namespace mdconv
