// hp_gemm2.hip -- GEMM-2 of the native 16-bit backward as a dense kernel (gfx950).
//
//     grad_W[o, c, tap] = sum_n grad_out[o, n] * col[tap][n][c]
//
// (reference: the `addmm_` over the column buffer, mdeformable_conv.cu:436-441; 3-D
// mdeformable_conv3d.cu:548-552), with col = the 16-bit column rows hp_bwd2_kernel leaves in the
// workspace next to the grad_col rows.  M = output channels, N = input channels, K = pixels.
// Until round 3 this contraction lived inside the fused backward kernel; its 64 accumulator
// registers per wave (cfg5) held that kernel at one workgroup per CU.  On its own it is a streaming
// kernel: per (tap, 32-pixel tile) a workgroup reads one grad_out tile (C_out x 32 x 2 B) and one
// column tile (32 x Cp x 2 B) -- 16 KB at cfg5 against 8 MFMAs per wave -- so it is HBM-bound by
// construction (cfg5 shard: 3.6 GB of column rows + the grad_out tiles of 27 taps, mostly cache
// hits, against 0.19 ms of matrix time) and is laid out for bytes in flight: 4-wave workgroups,
// about 100 registers, 4 workgroups per CU, next tile's loads issued before this tile's MFMAs, ONE
// barrier per tile (both LDS tiles double-buffered).
//
// Workgroup = (pixel range, tap); wave w = input channels [32w, 32w + 32); the accumulators of the
// whole range go to `part` ([tap][range][cblk][ob][lane][16], fp32) and hp_reduce_gw_kernel sums the
// ranges.  LDS tiles are written row-wise with 16-byte stores; the B operand (K = pixel, N = channel)
// is fetched from the [pixel][c] tile with ds_read_b64_tr_b16 exactly as the fused kernel did.
#include "hp_kernels.hpp"

namespace mdconv {

namespace {

template <typename T, int WAVES, int MB2>
__global__ __launch_bounds__(64 * WAVES, MB2 <= 4 ? 3 : 1) void hp_gemm2_kernel(
    Geom g, HpDims hd, const int4 *__restrict__ btab, const typename T::Raw *__restrict__ gout,
    const typename T::Raw *__restrict__ colbuf, float *__restrict__ part) {
  using Raw = typename T::Raw;
  constexpr int NT = 64 * WAVES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // column tile pitch Cp + 32 elements (16 dwords past a multiple of 64): the four rows a transposing
  // read touches per 16-lane group land 16 banks apart -- with Cp + 8 they were 4 banks apart and the
  // 8-byte pieces of neighbouring rows collided (1.6e8 conflict cycles per cfg5 launch)
  const int OpL = hd.OpL, Cp = hd.Cp, pitch = Cp + 32;
  Raw *Gop = reinterpret_cast<Raw *>(smem);     // [2][OpL][kPP]   grad_out tile, [o][pixel]
  Raw *Col = Gop + 2 * OpL * kPP;               // [2][32][pitch]  column tile,   [pixel][c]

  const int tid = threadIdx.x, lane = tid & 63, kh = lane >> 5, pl = lane & 31;
  const int cblk = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool active = cblk < hd.cblks;
  // unit = (range, tap), tap fastest, contiguous runs of units per XCD (xcd_remap): the K taps of a pixel
  // range run side by side on ONE XCD and read the same grad_out tiles -- from its L2 after the first
  const int unit = xcd_remap(blockIdx.x, gridDim.x);
  const int range = unit / g.K, tap = unit - range * g.K;
  const int t_lo = range * hd.tiles_per_range_w;
  const int t_hi = min(t_lo + hd.tiles_per_range_w, hd.ntiles);
  const int o_base = active ? btab[cblk].x : 0;

  f32x16 acc[MB2];
#pragma unroll
  for (int i = 0; i < MB2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  if (t_lo < t_hi) {
    // ---- loaders.  grad_out item = (o, pixel octet); column item = (pixel, channel octet).  (lb, lp) =
    // image / pixel of the first pixel of the tile being REQUESTED (wave-uniform) ----
    // tile_ok (32 | S_o): a tile never straddles two images, so a thread's items have CONSTANT offsets from
    // a per-tile scalar base -- buffer loads with the base in the scalar offset, no per-item address
    // arithmetic (the generic loaders below cost ~100 VALU per item against 8 MFMAs per tile and wave)
    const bool vec_ok = (g.S_o & 7) == 0, tile_ok = (g.S_o & 31) == 0;
    const rsrc_t r_go = make_rsrc(gout, (size_t)g.B * g.O * g.S_o * 2);
    const size_t col_img = (size_t)g.K * g.S_o * Cp;
    const int LPP = Cp / 8;
    const int ngitems = OpL * 4, ncitems = 32 * LPP;
    constexpr int GI = 2, CI = 2;   // items per thread kept in flight; more go through the tail loops
    int lb = (t_lo * 32) / g.S_o, lp = t_lo * 32 - lb * g.S_o;
    auto load_g = [&](int item, int lb, int lp) -> U4 {
      const int o = item >> 2, oct = item & 3;
      if (tile_ok)   // the scalar offset is NOT part of the range check: images beyond the batch are parked in the
                     // lane offset like padded channels (they cannot occur while tile_ok implies lb < B; kept explicit)
        return buf_load4u(r_go, o < g.O && lb < g.B ? (o * g.S_o + oct * 8) * 2 : kHpOob,
                          (min(lb, g.B - 1) * g.O * g.S_o + lp) * 2);
      int bb = lb, pp = lp + oct * 8;
      while (pp >= g.S_o) { pp -= g.S_o; ++bb; }
      U4 v = {0, 0, 0, 0};
      if (o < g.O && bb < g.B) {
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
      return v;
    };
    auto load_c = [&](int item, int lb, int lp) -> U4 {
      const int p = item / LPP, oc = item - p * LPP;
      if (tile_ok) {
        const rsrc_t r_c = make_rsrc(colbuf + (size_t)min(lb, g.B - 1) * col_img, lb < g.B ? col_img * 2 : 0);
        return buf_load4u_nt(r_c, (p * Cp + oc * 8) * 2, ((tap * g.S_o + lp) * Cp) * 2);
      }
      int bb = lb, pp = lp + p;
      while (pp >= g.S_o) { pp -= g.S_o; ++bb; }
      U4 v = {0, 0, 0, 0};   // pixels beyond the batch: the fused kernel never wrote those rows
      if (bb < g.B) v = *reinterpret_cast<const U4 *>(colbuf + (((int64_t)bb * g.K + tap) * g.S_o + pp) * Cp + oc * 8);
      return v;
    };
    auto store_g = [&](int item, const U4 &v, int buf) {
      *reinterpret_cast<U4 *>(Gop + (buf * OpL + (item >> 2)) * kPP + (item & 3) * 8) = v;
    };
    auto store_c = [&](int item, const U4 &v, int buf) {
      const int p = item / LPP, oc = item - p * LPP;
      *reinterpret_cast<U4 *>(Col + (buf * 32 + p) * pitch + oc * 8) = v;
    };
    // Two tiles ahead in registers (sets A and B), one more in the other LDS buffer: the kernel streams
    // 16 KB per tile and workgroup and is paced by bytes in flight, not by its 8 MFMAs per wave and tile.
    struct Regs { U4 g[GI], c[CI]; };
    int qb = lb, qp = lp;   // position of the tile being PUBLISHED (its tail items are loaded on the spot)
    auto next = [&](int &b_, int &p_) {
      p_ += 32;
      while (p_ >= g.S_o) { p_ -= g.S_o; ++b_; }
    };
    auto request = [&](Regs &r) {   // the tile at (lb, lp): first GI / CI items per thread into registers; advances
#pragma unroll
      for (int i = 0; i < GI; ++i) r.g[i] = tid + i * NT < ngitems ? load_g(tid + i * NT, lb, lp) : U4{0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < CI; ++i) r.c[i] = tid + i * NT < ncitems ? load_c(tid + i * NT, lb, lp) : U4{0, 0, 0, 0};
      next(lb, lp);
    };
    auto publish = [&](const Regs &r, int buf) {   // registers -> LDS (tile at (qb, qp)); advances
#pragma unroll
      for (int i = 0; i < GI; ++i) if (tid + i * NT < ngitems) store_g(tid + i * NT, r.g[i], buf);
#pragma unroll
      for (int i = 0; i < CI; ++i) if (tid + i * NT < ncitems) store_c(tid + i * NT, r.c[i], buf);
      for (int item = tid + GI * NT; item < ngitems; item += NT) store_g(item, load_g(item, qb, qp), buf);
      for (int item = tid + CI * NT; item < ncitems; item += NT) store_c(item, load_c(item, qb, qp), buf);
      next(qb, qp);
    };
    auto mma = [&](int buf) {
      if (active) {
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
          // B fragment (K = pixel, N = channel) from the [pixel][c] column tile
          U4 bc;
          lds_tr2(Col + (buf * 32 + ks2 * 16 + 8 * kh + ((lane & 15) >> 2)) * pitch + cblk * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3),
                  4 * pitch, bc);
#pragma unroll
          for (int ob = 0; ob < MB2; ++ob) {
            const U4 a = *reinterpret_cast<const U4 *>(Gop + (buf * OpL + o_base + ob * 32 + pl) * kPP + ks2 * 16 + 8 * kh);
            acc[ob] = T::mfma(a, bc, acc[ob]);
          }
        }
      }
    };

    Regs ra, rb;
    request(ra);
    publish(ra, 0);
    if (t_lo + 1 < t_hi) request(ra);
    if (t_lo + 2 < t_hi) request(rb);
    __syncthreads();
    // iteration `tile`: tile + 1 goes to the other LDS buffer (last read before the barrier that ended the
    // previous iteration), tile + 3 is requested into the register set that just emptied
    for (int tile = t_lo; tile < t_hi; tile += 2) {
      if (tile + 1 < t_hi) {
        publish(ra, 1);
        if (tile + 3 < t_hi) request(ra);
      }
      mma(0);
      __syncthreads();
      if (tile + 1 < t_hi) {
        if (tile + 2 < t_hi) {
          publish(rb, 0);
          if (tile + 4 < t_hi) request(rb);
        }
        mma(1);
        __syncthreads();
      }
    }
  }
  if (active) {
    float4 *dst = reinterpret_cast<float4 *>(
        part + ((((int64_t)tap * hd.ranges_w + range) * hd.cblks + cblk) * MB2) * 1024 + lane * 16);
#pragma unroll
    for (int ob = 0; ob < MB2; ++ob)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        dst[ob * 256 + q] = make_float4(acc[ob][4 * q], acc[ob][4 * q + 1], acc[ob][4 * q + 2], acc[ob][4 * q + 3]);
  }
}

template <typename T, int WAVES, int MB2>
int launch_gemm2(const Geom &g, const HpDims &hd, const Tensors &t, const int4 *btab, const void *colbuf,
                 float *part, hipStream_t stream) {
  using Raw = typename T::Raw;
  const size_t lds = hp_gemm2_lds_bytes(hd);
  if (lds > 64 * 1024) {
    hipError_t ea = hipFuncSetAttribute((const void *)hp_gemm2_kernel<T, WAVES, MB2>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ea != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(ea)); return MDCONV_ELAUNCH; }
  }
  hp_debug_plan("hp_gemm2", hp_gemm2_kernel<T, WAVES, MB2>, 64 * WAVES, lds, (long)hd.ranges_w * g.K);
  hipLaunchKernelGGL((hp_gemm2_kernel<T, WAVES, MB2>), dim3(hd.ranges_w * g.K), dim3(64 * WAVES), lds, stream, g, hd,
                     btab, (const Raw *)t.grad_output, (const Raw *)colbuf, part);
  return check_launch("hp_gemm2");
}

template <typename T>
int dispatch_gemm2(const Geom &g, const HpDims &hd, const Tensors &t, const int4 *btab, const void *colbuf,
                   float *part, hipStream_t stream) {
#define HP_G2(W)                                                                       \
  switch (hd.MB2) {                                                                    \
    case 1: return launch_gemm2<T, W, 1>(g, hd, t, btab, colbuf, part, stream);        \
    case 2: return launch_gemm2<T, W, 2>(g, hd, t, btab, colbuf, part, stream);        \
    case 4: return launch_gemm2<T, W, 4>(g, hd, t, btab, colbuf, part, stream);        \
    default: return launch_gemm2<T, W, 8>(g, hd, t, btab, colbuf, part, stream);       \
  }
  switch (hd.waves) {
    case 1: HP_G2(1);
    case 2: HP_G2(2);
    case 4: HP_G2(4);
    default: HP_G2(8);
  }
#undef HP_G2
}

}  // namespace

size_t hp_gemm2_lds_bytes(const HpDims &hd) {
  return (size_t)2 * hd.OpL * kPP * 2 + (size_t)2 * 32 * (hd.Cp + 32) * 2;
}

int hp_gemm2_launch(const Geom &g, const HpDims &hd, int dtype, const Tensors &t, const int4 *btab,
                    const void *colbuf, float *part, hipStream_t stream) {
  if (dtype == MDCONV_F16) return dispatch_gemm2<F16>(g, hd, t, btab, colbuf, part, stream);
  return dispatch_gemm2<BF16>(g, hd, t, btab, colbuf, part, stream);
}

}  // namespace mdconv
