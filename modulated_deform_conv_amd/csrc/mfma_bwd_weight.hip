// mfma_bwd_weight.hip -- grad_weight (+ grad_bias) as a fused implicit GEMM (fp32, gfx950).
//
//   grad_W[o, (tap, c)] += sum_n grad_out[o, n] * col[(tap, c), n],   n = flattened (b, pix)
//
// the reference's GEMM-2 (mdeformable_conv.cu:436-439) with the `columns` operand re-gathered on
// the fly instead of being re-materialised by the gradient kernel (mdeformable_conv.cu:316).
// M = output channels, N = 32 input channels of one tap, K = pixels, split over `splits`
// workgroups whose partial tiles are summed by a small reduction kernel.
//
// Same VALU-starved structure as the forward kernel (see mfma_fwd.hip): both operands arrive by
// raw buffer loads whose addresses live in SGPRs / lane constants:
//   * A = grad_out in MFMA-fragment order, emitted by GEMM-1 (mfma_bwd_data.hip, `emit_ga`);
//   * B = col slab [16 pixels][32 channels]: a thread owns (pixel kk, channels sub, sub+16); the
//     sampling state of (tap, pixel) changes every chunk, so it is NOT recomputed here (that
//     would be ~60 VALU per chunk) but read from the per-call `tap table` (byte offsets + weights
//     of the 2^ND corners, built by `build_tap_table`), two chunks ahead;
//     (gathering two chunks ahead with double-buffered corner registers was measured SLOWER:
//     1.14 -> 1.24 ms at cfg2);
//   * LDS holds only the B slab (double buffered, pitch 33 so the transposing write is at most
//     2-way conflicted, which is free for ds_write_b32).
#include "mfma_kernels.hpp"
#include "mfma_tile.hpp"

namespace mdconv {

namespace {


// ---------------------------------------------------------------------------------------------
// tap table: per (dg, tap, n) the byte offsets of the 2^(ND-1) corner PAIRS (image base folded
// in; the two neighbours along the contiguous axis are fetched by one 8-byte load) and the 2^ND
// weights (validity, backward load gating and the mask folded in).
// Entry = kTabWords(ND) words: [NP offsets | pad][NC weights]  (8 words in 2-D, 16 in 3-D).
// ---------------------------------------------------------------------------------------------
// (The table is written by GEMM-1, mfma_bwd_data.hip `new_tap_state`, which computes the same
// sampling state anyway.)
// ---------------------------------------------------------------------------------------------
// the GEMM
// ---------------------------------------------------------------------------------------------
// Wave arrangement: WR x WC waves, MB 32-row tiles per wave -> workgroup tile
// (WR * MB * 32 output channels) x (WC * 32 input channels):
//   <4, 1, 2>  256 x 32   (a 64 x 64 arrangement for C_out <= 64 measured slower with NCHW gathers -- GEMM-2 5.4 -> 8 ms at
//                          cfg4, round 1 -- and is not instantiated any more; narrow shapes run the channels-last kernels).
template <int ND, bool PADN, int WR, int WC, int MB>
__global__ __launch_bounds__(256) void mfma_bwd_weight_kernel(Geom g, BwdDims bd,
                                                              const float *__restrict__ input,
                                                              const float *__restrict__ ga,
                                                              const int *__restrict__ table,
                                                              float *__restrict__ part) {
  static_assert(WR * WC == 4, "four waves");
  constexpr int NC = 1 << ND, NP = NC / 2;
  constexpr int BK = kBK;
  constexpr int RM = WR * MB * 32, CN = WC * 32;   // rows / channels per workgroup
  constexpr int CT = CN / 16;                       // channels gathered per thread per chunk
  constexpr int kPitch = CN + 1;
  __shared__ __attribute__((aligned(16))) float Bs[2 * BK * kPitch];

  // blockIdx.x = (mtile * K + tap) * cblks + cblk ; blockIdx.y = split
  int id = blockIdx.x;
  const int cblk = id % bd.cblks; id /= bd.cblks;
  const int tap = id % g.K;
  const int mtile = id / g.K;
  const int split = blockIdx.y;
  const int c0 = cblk * CN;
  const int dg = min(c0, g.C - 1) / g.Cdg;

  const int tid = threadIdx.x, lane = tid & 63, kh = lane >> 5;
  // wave id as an SGPR: anything derived from threadIdx is 'divergent' to hipcc, and a divergent
  // buffer soffset is wrapped in a readfirstlane waterfall per load (cdna_hip_programming.md T20)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wcn = wave % WC;
  const int kk = tid & 15, sub = tid >> 4;   // pixel within the chunk, channel within the tile

  const int pairs_total = bd.Np / 32;
  const int p_begin = split * bd.pairs_per_split;
  const int p_end = min(p_begin + bd.pairs_per_split, pairs_total);
  const int t_begin = 2 * p_begin, t_end = 2 * p_end;   // chunk range (16 pixels each), even count

  const rsrc_t r_in = make_rsrc(input, (size_t)g.B * g.C * g.S_i * sizeof(float));
  const int slab_bytes = bd.mblks * 2 * 64 * 16;
  const rsrc_t r_ga = make_rsrc(ga, (size_t)(bd.Np / 16) * slab_bytes);
  const int entry_bytes = 2 * NC * 4;
  const rsrc_t r_tab = make_rsrc(table + (size_t)(dg * g.K + tap) * bd.Np * (2 * NC),
                                 (size_t)bd.Np * entry_bytes);
  const int wo = mtile * RM + wr * MB * 32;   // first output channel of this wave
  const int a_voff = ((wo / 32) * 2 * 64 + lane) * 16;
  // rows of this wave beyond C_out are padding: skip their fragment loads and MFMAs (the wave
  // still gathers and synchronises)
  // ... and with conv groups only the output channels of the groups that own this wave's 32
  // input channels can receive a gradient (the dense product is block diagonal)
  bool m_active = wo < g.O;
  if (g.G > 1) {
    const int cw = c0 + wcn * 32;
    const int o_lo = (min(cw, g.C - 1) / g.Cg) * g.Og, o_hi = (min(cw + 31, g.C - 1) / g.Cg + 1) * g.Og;
    m_active = m_active && wo < o_hi && wo + MB * 32 > o_lo;
  }
  const int t_voff = kk * entry_bytes;
  const int chan_voff = min(c0 + sub, g.C - 1) * g.S_i * 4;   // this thread's first channel plane
  const int chan_soff = 16 * g.S_i * 4;                       // the others are 16 planes apart

  f32x16 acc[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // vo: byte offsets of the corner pairs; their two low bits flag an element that comes along with the 8-byte load
  // but that the reference never reads (bit 0: the first, bit 1: the second; written by GEMM-1) -- its weight is 0,
  // and it must not be multiplied at all (0 * Inf = NaN): `commit` selects it away behind a wave-uniform branch
  struct Tab { int vo[NP]; float w[NC]; };
  auto load_tab = [&](Tab &tb, int t) {
    const int soff = t * 16 * entry_bytes;
    if constexpr (ND == 2) {
      const float4 a = buf_load4(r_tab, t_voff, soff);            // [vo0 vo1 - -]
      tb.vo[0] = __float_as_int(a.x); tb.vo[1] = __float_as_int(a.y);
      const float4 b = buf_load4(r_tab, t_voff + 16, soff);
      tb.w[0] = b.x; tb.w[1] = b.y; tb.w[2] = b.z; tb.w[3] = b.w;
    } else {
      const float4 a = buf_load4(r_tab, t_voff, soff);            // [vo0..vo3]
      tb.vo[0] = __float_as_int(a.x); tb.vo[1] = __float_as_int(a.y);
      tb.vo[2] = __float_as_int(a.z); tb.vo[3] = __float_as_int(a.w);
#pragma unroll
      for (int h = 0; h < NC / 4; ++h) {
        const float4 b = buf_load4(r_tab, t_voff + NC * 4 + h * 16, soff);
        tb.w[4 * h + 0] = b.x; tb.w[4 * h + 1] = b.y; tb.w[4 * h + 2] = b.z; tb.w[4 * h + 3] = b.w;
      }
    }
  };
  float2 rg[CT][NP];
  auto gather = [&](const Tab &tb) {
#pragma unroll
    for (int pi = 0; pi < NP; ++pi) {
      // unsigned arithmetic: a parked pair (0x7ffffff0, written by GEMM-1) plus the channel plane wraps past 2^31, which
      // the buffer's bounds check reads as an out-of-range UNSIGNED offset (loads give 0) -- every per-chunk tensor stays
      // below 2 GiB (chunk_limit(), mfma_kernels.hip), so no in-range sum can reach that value
      const int vo = (int)(((unsigned)tb.vo[pi] & ~3u) + (unsigned)chan_voff);
#pragma unroll
      for (int i = 0; i < CT; ++i) rg[i][pi] = buf_load2(r_in, vo, i * chan_soff);
    }
  };
  auto commit = [&](const Tab &tb, int t, float *Bb) {
    int flags = 0;
#pragma unroll
    for (int pi = 0; pi < NP; ++pi) flags |= tb.vo[pi];
    if (__any((flags & 3) != 0)) {   // wave-uniform
#pragma unroll
      for (int pi = 0; pi < NP; ++pi)
#pragma unroll
        for (int i = 0; i < CT; ++i) {
          rg[i][pi].x = (tb.vo[pi] & 1) ? 0.f : rg[i][pi].x;
          rg[i][pi].y = (tb.vo[pi] & 2) ? 0.f : rg[i][pi].y;
        }
    }
#pragma unroll
    for (int i = 0; i < CT; ++i) {
      float val = tb.w[0] * rg[i][0].x;
      val = fmaf(tb.w[1], rg[i][0].y, val);
#pragma unroll
      for (int pi = 1; pi < NP; ++pi) {
        val = fmaf(tb.w[2 * pi], rg[i][pi].x, val);
        val = fmaf(tb.w[2 * pi + 1], rg[i][pi].y, val);
      }
      if (PADN) val = (t * 16 + kk < g.N) ? val : 0.f;
      Bb[kk * kPitch + sub + 16 * i] = val;
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
    const float *Bb = Bbuf + wcn * 32 + (lane & 31) + 4 * kh * kPitch;
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
  float *dst = part + ((size_t)(split * g.K + tap) * bd.OgpB) * bd.Cp + c0 + wcn * 32 + (lane & 31);
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = wo + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      dst[(size_t)o * bd.Cp] = acc[mb][r];
    }
}

// grad_weight[o][c_local][tap] += sum_split part[split][tap][o][c], c = group(o) * Cg + c_local
// (only the block-diagonal entries of the dense product exist in the grouped weight)
__global__ __launch_bounds__(256) void reduce_weight_kernel(Geom g, BwdDims bd,
                                                            const float *__restrict__ part,
                                                            float *__restrict__ grad_weight) {
  const int64_t total = (int64_t)g.K * g.O * g.Cg;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cl = (int)(i % g.Cg);
    const int o = (int)((i / g.Cg) % g.O);
    const int tap = (int)(i / g.Cg / g.O);
    const int c = (o / g.Og) * g.Cg + cl;
    // four partials in flight (a plain loop left one dependent load per split: 30 us for 33 MB at cfg2); the order
    // of the additions is fixed, so the result stays bit-reproducible
    const float *src = part + ((size_t)tap * bd.OgpB + o) * bd.Cp + c;
    const size_t stride = (size_t)g.K * bd.OgpB * bd.Cp;
    float s = 0.f;
    int sp = 0;
    for (; sp + 8 <= bd.splits; sp += 8) {   // eight partials in flight (beside the forked gather a load takes microseconds)
      float a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = src[(size_t)(sp + u) * stride];
      s += ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    for (; sp + 4 <= bd.splits; sp += 4) {
      const float a0 = src[(size_t)sp * stride], a1 = src[(size_t)(sp + 1) * stride];
      const float a2 = src[(size_t)(sp + 2) * stride], a3 = src[(size_t)(sp + 3) * stride];
      s += (a0 + a1) + (a2 + a3);
    }
    for (; sp < bd.splits; ++sp) s += src[(size_t)sp * stride];
    float *dst = grad_weight + ((int64_t)o * g.Cg + cl) * g.K + tap;
    *dst = g.acc_w ? *dst + s : s;
  }
}

// grad_bias[o] += sum_{b, pix} grad_out[b][o][pix]   (mdeformable_conv.cu:440-444): GEMM-1 leaves one
// partial sum per (pixel tile, o) (mfma_bwd_data.hip, emit_ga); they are added here in a fixed
// order (two small stages) -- no floating-point atomics, the result is bit-identical from run to
// run.
constexpr int kBiasSlices = 32;
// stage 1: grid (O / 64, kBiasSlices); thread = (o, quarter): tiles k = slice * 4 + quarter (mod 128)
__global__ __launch_bounds__(256) void grad_bias_stage1_kernel(Geom g, int tiles,
                                                               const float *__restrict__ partial,
                                                               float *__restrict__ stage) {
  __shared__ float red[4][64];
  const int o = blockIdx.x * 64 + (threadIdx.x & 63);
  const int quarter = threadIdx.x >> 6;
  float s = 0.f;
  if (o < g.O) {
    // four partials in flight, added in a fixed order (a plain loop waited for one load per addition: 27 us at cfg2)
    constexpr int kStep = 4 * kBiasSlices;
    int k = blockIdx.y * 4 + quarter;
    for (; k + 3 * kStep < tiles; k += 4 * kStep) {
      const float a0 = partial[(size_t)k * g.O + o], a1 = partial[(size_t)(k + kStep) * g.O + o];
      const float a2 = partial[(size_t)(k + 2 * kStep) * g.O + o], a3 = partial[(size_t)(k + 3 * kStep) * g.O + o];
      s += (a0 + a1) + (a2 + a3);
    }
    for (; k < tiles; k += kStep) s += partial[(size_t)k * g.O + o];
  }
  red[quarter][threadIdx.x & 63] = s;
  __syncthreads();
  if (quarter == 0 && o < g.O)
    stage[(size_t)blockIdx.y * g.O + o] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// stage 2: one ordered sum of the kBiasSlices stage values per output channel
__global__ __launch_bounds__(256) void grad_bias_final_kernel(Geom g, const float *__restrict__ stage,
                                                              float *__restrict__ grad_bias) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= g.O) return;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < kBiasSlices; ++k) s += stage[(size_t)k * g.O + o];
  grad_bias[o] = g.acc_w ? grad_bias[o] + s : s;
}

int grid_for(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace

int mfma_bwd_weight_occupancy(int nd, bool padn, int wtile) {
  static int cache[2][2][2] = {};
  int &slot = cache[nd == 3][padn][wtile == 1];
  if (slot) return slot;
  int n = 0;
#define OCC_BW(ND, PADN, WR, WC, MB)                                                                           \
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(                                                          \
      &n, reinterpret_cast<const void *>(&mfma_bwd_weight_kernel<ND, PADN, WR, WC, MB>), 256, 0)
#define OCC_BW2(ND, PADN) OCC_BW(ND, PADN, 4, 1, 2)
  if (nd == 2) { if (padn) OCC_BW2(2, true); else OCC_BW2(2, false); }
  else { if (padn) OCC_BW2(3, true); else OCC_BW2(3, false); }
#undef OCC_BW2
#undef OCC_BW
  (void)hipGetLastError();
  if (n <= 0) n = 4;
  slot = n;
  return n;
}

int mfma_bwd_weight_f32(const Geom &g, const BwdDims &bd, const Tensors &t, const float *ga,
                        const int *table, float *part, const float *bias_part, const float *xt,
                        hipStream_t stream) {
  const dim3 grid(bd.mtiles * g.K * bd.cblks, bd.splits);
  const bool padn = bd.Np != g.N;
  profile_mark(2, true, stream, bd.cl ? "mfma_bwd_weight_cl_kernel" : "mfma_bwd_weight_kernel");
  if (bd.cl) {
    const int rcl = mfma_bwd_weight_cl_launch(g, bd, xt, ga, table, part, stream);
    if (rcl) return rcl;
  } else {
#define LAUNCH_BW(ND, PADN, WR, WC, MB)                                                         \
  hipLaunchKernelGGL((mfma_bwd_weight_kernel<ND, PADN, WR, WC, MB>), grid, dim3(256), 0, stream, \
                     g, bd, (const float *)t.input, ga, table, part)
#define LAUNCH_BW2(ND, PADN) LAUNCH_BW(ND, PADN, 4, 1, 2)
  if (g.nd == 2) { if (padn) LAUNCH_BW2(2, true); else LAUNCH_BW2(2, false); }
  else { if (padn) LAUNCH_BW2(3, true); else LAUNCH_BW2(3, false); }
#undef LAUNCH_BW2
#undef LAUNCH_BW
  }
  profile_mark(2, false, stream);
  int rc = check_launch("mfma_bwd_weight");
  if (rc) return rc;
  hipLaunchKernelGGL(reduce_weight_kernel, dim3(grid_for((int64_t)g.K * g.O * g.Cg)), dim3(256), 0,
                     stream, g, bd, part, (float *)t.grad_weight);
  return check_launch("reduce_weight");
}

// grad_bias: its inputs (GEMM-1's per-tile partial sums) are ready before GEMM-2 starts, so the caller runs it
// beside GEMM-2 on the forked stream instead of behind the split-K reduction (two launches and their gaps off the
// critical path); own scratch for the stage values (it used to borrow GEMM-2's partial buffer)
size_t grad_bias_stage_bytes(const Geom &g) { return (size_t)kBiasSlices * g.O * sizeof(float); }

int grad_bias_f32(const Geom &g, const BwdDims &bd, const float *bias_part, float *stage, float *grad_bias,
                  hipStream_t stream) {
  if (!g.with_bias) return MDCONV_OK;
  hipLaunchKernelGGL(grad_bias_stage1_kernel, dim3((g.O + 63) / 64, kBiasSlices), dim3(256), 0, stream, g,
                     bd.bias_tiles, bias_part, stage);
  int rc = check_launch("grad_bias_stage1");
  if (rc) return rc;
  hipLaunchKernelGGL(grad_bias_final_kernel, dim3((g.O + 255) / 256), dim3(256), 0, stream, g, stage, grad_bias);
  return check_launch("grad_bias_final");
}

}  // namespace mdconv
