// hp_bwd2.hip -- fused backward kernel of the native 16-bit path, line-wide gathers (gfx950).
//
// Same mathematics and outputs as hp_bwd.hip (reference: mdeformable_conv.cu:412-444, 202-318; 3-D
// mdeformable_conv3d.cu:515-560, 265-395): per (tap, 32-pixel tile)
//     GEMM-1  gc[c, n] = sum_o W[o, c, tap] grad_out[o, n]
//     S[ci]   = sum_c gc[c] x[ci][c]  -> grad_mask, grad_offset;   grad_col row -> workspace
//     col[c]  = mask * sum_ci w[ci] x[ci][c]
//     GEMM-2  grad_W[o, c] += sum_n grad_out[o, n] col[n, c]
// with ONE gather of every corner.  What changed is WHO gathers: measured on MI355X
// (tools/ubench_gather16.hip) a 16-byte-per-lane load runs at full texture-path rate only when
// aligned quads of lanes read 64 contiguous bytes; the MFMA-native mapping of hp_bwd.hip (lane =
// pixel) is 4x slower.  So the work of a tile is done in three phases with different thread roles
// and LDS hand-overs between them (2 barriers per tile, grad_out tiles double-buffered):
// LDS tiles are only ever WRITTEN row-wise with 16-byte stores; where a matrix operand needs the
// other orientation it is fetched with ds_read_b64_tr_b16 (gfx950's transposing LDS read: the 16
// lanes of a group address the 8-byte pieces of a 4 x 16 block, lane i receives column i).
//   P2  wave = 32 input channels:  GEMM-1 from the grad_out tile in LDS, result -> LDS Gc[pixel][c]
//   P3  thread = (pixel, channel octet), Cp/8 ADJACENT lanes per pixel: reads its 16-byte piece of
//       the grad_col row from Gc and streams it to the workspace (whole rows, coalesced), gathers
//       its octet of every corner (a pixel's corner = Cp*2 contiguous bytes), accumulates S[ci]
//       (v_dot2c) and col (v_fma_mix), writes its col piece to LDS, reduces S over the lanes of its
//       (pixel, deformable group) with shuffles
//   P4  wave = 32 input channels:  GEMM-2 from the column / grad_out tiles; threads
//       (pixel, deformable group) finish grad_offset / grad_mask (single owner, no atomics), then
//       build the sampling state of the NEXT tile into the LDS state table (+ CSR counting)
#include "hp_kernels.hpp"

namespace mdconv {

namespace {



// advance (b, oc[]) -- image index and output coordinates of a pixel -- by `adv` flattened pixels
template <int ND> __device__ __forceinline__ void advance_pixel(const Geom &g, int adv, int &b, int *oc) {
  oc[ND - 1] += adv;
#pragma unroll
  for (int a = ND - 1; a > 0; --a)
    while (oc[a] >= g.out_sz[a]) { oc[a] -= g.out_sz[a]; ++oc[a - 1]; }
  while (oc[0] >= g.out_sz[0]) { oc[0] -= g.out_sz[0]; ++b; }
}

// Workgroup = WAVES worker waves (wave w = input channels [32w, 32w+32) in the matrix phases, the
// (pixel, octet) items in the gather phase) + NS state waves (two when DG > 2, i.e. more than 64
// (pixel, deformable group) states per tile: the state pipeline is the critical path there) that run the per-(tap, pixel) scalar
// pipeline -- offsets / mask, sampling state, CSR counting, the final grad_offset / grad_mask
// arithmetic -- beside them (it was 47 % of the tile time when wave 0 did it on top of its share).
template <int ND, bool MOD, typename T, int WAVES, int NKS, int NS>
__global__ __launch_bounds__(64 * (WAVES + NS), WAVES >= 8 ? 1 : 2) void hp_bwd2_kernel(
    Geom g, HpDims hd, const typename T::Raw *__restrict__ xt, const U4 *__restrict__ wpb,
    const int4 *__restrict__ btab, const typename T::Raw *__restrict__ gout,
    const typename T::Raw *__restrict__ offset, const typename T::Raw *__restrict__ mask,
    typename T::Raw *__restrict__ gcol, typename T::Raw *__restrict__ grad_offset,
    typename T::Raw *__restrict__ grad_mask, float *__restrict__ part, int *__restrict__ cnt) {
  using Raw = typename T::Raw;
  constexpr int NC = 1 << ND, NP = NC / 2;
  constexpr int MB2 = NKS / 2;
  constexpr int NTW = 64 * WAVES;   // worker threads
  constexpr int SW = 2 * NC + 4;    // state dwords per (pixel, dg): voff[NC], w*mask[NC], grad_col row, pad
  constexpr int SP = 2 / NS;        // (pixel, dg) states per lane of a state wave (DG <= 4; NS = 2 state waves when DG > 2)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int OpL = hd.OpL, Cp = hd.Cp;
  const int pitch_gc = Cp + 8;
  Raw *Gop = reinterpret_cast<Raw *>(smem);            // [2][OpL][kPP]   grad_out tile, [o][pixel]
  Raw *Gc = Gop + 2 * OpL * kPP;                       // [32][pitch_gc]  grad_col tile, [pixel][c]
  Raw *Col = Gc + 32 * pitch_gc;                       // [32][pitch_gc]  column tile,   [pixel][c]
  int *St = reinterpret_cast<int *>(Col + 32 * pitch_gc);              // [2][32 * DG][SW]
  float *Spart = reinterpret_cast<float *>(St + 2 * 32 * g.DG * SW);   // [32 * DG][msub][NC]

  const int tid = threadIdx.x, lane = tid & 63, kh = lane >> 5, pl = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_state = wave >= WAVES;
  const int cblk = wave;
  const bool active = !is_state && cblk < hd.cblks;
  const int tap = blockIdx.y, range = blockIdx.x;
  const int t_lo = range * hd.tiles_per_range;
  const int t_hi = min(t_lo + hd.tiles_per_range, hd.ntiles);
  if (t_lo >= t_hi) return;

  const int S_e = hp_anchor_space(g);              // scatter anchors per (image, deformable group)
  const int LPP = Cp / 8;                          // lanes per pixel in the gather phase (multiple of 4)
  const int LPD = g.DG == 1 ? LPP : g.Cdg / 8;     // lanes per (pixel, deformable group)
  int sub = 1;                                     // shuffle-reduced lanes: largest power of two | LPD, <= 64
  while (sub < 64 && LPD % (sub * 2) == 0) sub *= 2;
  const int msub = LPD / sub;                      // partial sums left for the owner thread

  // first pixel of the range: (image, pixel in image); every later position is tracked incrementally
  const int n_first = t_lo * 32;
  const int b_first = n_first / g.S_o, p_first = n_first - b_first * g.S_o;


  if (is_state) {
    // =====================================================================================
    // state wave
    // =====================================================================================
    const int npass = (32 * g.DG + 64 * NS - 1) / (64 * NS);
    int tcd[ND];
    tap_coords<ND>(g, tap, tcd);
    // per-lane (pixel, dg) items; position of the pixel whose state is built NEXT
    bool on[SP];
    int dgi[SP], nb[SP], noc[SP][ND];
#pragma unroll
    for (int ps = 0; ps < SP; ++ps) {
      const int x = lane + 64 * (NS == 2 ? wave - WAVES : ps);
      on[ps] = ps < npass && x < 32 * g.DG;
      dgi[ps] = on[ps] ? x >> 5 : 0;
      nb[ps] = b_first;
      out_coords<ND>(g, p_first, noc[ps]);
      advance_pixel<ND>(g, pl, nb[ps], noc[ps]);
    }
    Raw dlr[SP][ND], mlr[SP];   // raw 16-bit values until build() (a conversion in fetch() is a use of the load where it is issued)
    auto fetch = [&]() {   // offsets / mask of the pixel at (nb, noc)
#pragma unroll
      for (int ps = 0; ps < SP; ++ps) {
        if (on[ps]) {
          const int bb = min(nb[ps], g.B - 1);
          int pix = noc[ps][0];
#pragma unroll
          for (int a = 1; a < ND; ++a) pix = pix * g.out_sz[a] + noc[ps][a];
          const int64_t seg = (int64_t)bb * g.DG + dgi[ps];
          const int64_t ob = (seg * (ND * g.K) + ND * tap) * g.S_o + pix;
#pragma unroll
          for (int a = 0; a < ND; ++a) dlr[ps][a] = offset[ob + (int64_t)a * g.S_o];
          if (MOD) mlr[ps] = mask[(seg * g.K + tap) * g.S_o + pix];
        }
      }
    };
    struct Fac { float wl[ND], wh[ND], sl[ND], sh[ND], mg, old[ND + 1]; int64_t off_idx, msk_idx; bool live; };
    Fac cur[SP], nxt[SP];
#pragma unroll
    for (int ps = 0; ps < SP; ++ps) { cur[ps].live = false; nxt[ps].live = false; }
    auto build = [&](int slot) {   // state of the pixel at (nb, noc) from dl / ml -> LDS table; then advance
#pragma unroll
      for (int ps = 0; ps < SP; ++ps) {
        if (on[ps]) {
          Fac &f = nxt[ps];
          f.live = nb[ps] < g.B;
          const int bb = min(nb[ps], g.B - 1);
          int pix = noc[ps][0];
#pragma unroll
          for (int a = 1; a < ND; ++a) pix = pix * g.out_sz[a] + noc[ps][a];
          float dlf[ND], mlf = 1.f;
#pragma unroll
          for (int a = 0; a < ND; ++a) dlf[a] = T::ldf(&dlr[ps][a]);
          if (MOD) mlf = T::ldf(&mlr[ps]);
          TapCoef<ND, float> tc;
          make_tap<ND, float>(g, noc[ps], tcd, dlf, true, tc);
          HpCorners<ND> hc;
          hp_corners<ND>(tc, hc);
          f.mg = (!g.range_gate || tc.inside) ? mlf : 0.f;
#pragma unroll
          for (int a = 0; a < ND; ++a) { f.wl[a] = tc.wl[a]; f.wh[a] = tc.wh[a]; f.sl[a] = tc.sl[a]; f.sh[a] = tc.sh[a]; }
          int ev[SW];
#pragma unroll
          for (int ci = 0; ci < NC; ++ci) {
            ev[ci] = (f.live && hc.idx[ci] >= 0) ? (bb * g.S_i + hc.idx[ci]) * Cp * 2 : kHpOob;
            ev[NC + ci] = __float_as_int(f.live ? hc.w[ci] * mlf : 0.f);
          }
          // grad_col row: byte offset inside its image's rows (one image's rows stay below the chunk
          // limit, a whole chunk's need not), and the image
          ev[2 * NC] = f.live ? (tap * g.S_o + pix) * Cp * 2 : kHpOob;
          ev[2 * NC + 1] = bb;
          ev[2 * NC + 2] = ev[2 * NC + 3] = 0;
          int *sp = St + ((slot * 32 + pl) * g.DG + dgi[ps]) * SW;
#pragma unroll
          for (int q = 0; q < SW; q += 4) *reinterpret_cast<int4 *>(sp + q) = make_int4(ev[q], ev[q + 1], ev[q + 2], ev[q + 3]);
          const int64_t seg = (int64_t)bb * g.DG + dgi[ps];
          f.off_idx = (seg * (ND * g.K) + ND * tap) * g.S_o + pix;
          f.msk_idx = (seg * g.K + tap) * g.S_o + pix;
#pragma unroll
          for (int a = 0; a <= ND; ++a) f.old[a] = 0.f;
          if (f.live && g.acc_data) {   // accumulate mode: previous values, needed a tile later
#pragma unroll
            for (int a = 0; a < ND; ++a) f.old[a] = T::ldf(grad_offset + f.off_idx + (int64_t)a * g.S_o);
            if (MOD) f.old[ND] = T::ldf(grad_mask + f.msk_idx);
          }
          if (f.live) {
            // scatter anchor of this sample (first pass of the CSR build, hp_col2im.hip): one
            // fire-and-forget integer atomic per sample
            SampleAnchor<ND> sa;
            sample_anchor<ND>(g, tc, 1.f, sa);
            if (sa.on) atomicAdd(cnt + seg * S_e + sa.qa, 1);
          }
          advance_pixel<ND>(g, 32, nb[ps], noc[ps]);
        }
      }
    };
    auto finish = [&]() {   // grad_offset / grad_mask of the tile whose factors are in `cur`
#pragma unroll
      for (int ps = 0; ps < SP; ++ps) {
        const Fac &f = cur[ps];
        if (on[ps] && f.live) {
          float S[NC];
          const float *sp = Spart + ((pl * g.DG + dgi[ps]) * msub) * NC;
#pragma unroll
          for (int ci = 0; ci < NC; ++ci) S[ci] = sp[ci];
          for (int mi = 1; mi < msub; ++mi)
#pragma unroll
            for (int ci = 0; ci < NC; ++ci) S[ci] += sp[mi * NC + ci];
          float gm = 0.f, goff[ND];
#pragma unroll
          for (int a = 0; a < ND; ++a) goff[a] = 0.f;
#pragma unroll
          for (int ci = 0; ci < NC; ++ci) {
            float w = 1.f;
#pragma unroll
            for (int a = 0; a < ND; ++a) w *= ((ci >> (ND - 1 - a)) & 1) ? f.wh[a] : f.wl[a];
            gm = fmaf(w, S[ci], gm);
#pragma unroll
            for (int a = 0; a < ND; ++a) {
              float dw = 1.f;
#pragma unroll
              for (int a2 = 0; a2 < ND; ++a2) {
                const bool hi = (ci >> (ND - 1 - a2)) & 1;
                dw *= (a2 == a) ? (hi ? f.sh[a2] : f.sl[a2]) : (hi ? f.wh[a2] : f.wl[a2]);
              }
              goff[a] = fmaf(dw, S[ci], goff[a]);
            }
          }
#pragma unroll
          for (int a = 0; a < ND; ++a) T::stf(grad_offset + f.off_idx + (int64_t)a * g.S_o, goff[a] * f.mg + f.old[a]);
          if (MOD) T::stf(grad_mask + f.msk_idx, gm + f.old[ND]);
        }
      }
    };

    fetch();
    build(0);
    if (t_lo + 1 < t_hi) fetch();
    __syncthreads();   // prologue barrier
    for (int tile = t_lo; tile < t_hi; ++tile) {
      if (tile > t_lo) finish();
#pragma unroll
      for (int ps = 0; ps < SP; ++ps) cur[ps] = nxt[ps];
      __syncthreads();   // B2
      if (tile + 1 < t_hi) {
        build((tile + 1 - t_lo) & 1);
        if (tile + 2 < t_hi) fetch();
      }
      __syncthreads();   // B3
    }
    finish();
    return;
  }

  // =======================================================================================
  // worker waves
  // =======================================================================================
  const int o_base = active ? btab[cblk].x : 0;
  // gather role: item = (pixel, channel octet); LPP adjacent lanes per pixel
  const int nitem = 32 * LPP;
  int it_p[2], it_oc[2], it_dg[2];
  bool it_on[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int item = tid + k * NTW;
    it_on[k] = item < nitem;
    it_p[k] = it_on[k] ? item / LPP : 0;
    it_oc[k] = it_on[k] ? item - it_p[k] * LPP : 0;
    it_dg[k] = g.DG == 1 ? 0 : it_oc[k] / LPD;
  }
  const rsrc_t r_xt = make_rsrc(xt, (size_t)g.B * g.S_i * Cp * 2);
  const size_t gcol_img = (size_t)g.K * g.S_o * Cp;   // grad_col elements per image
  const rsrc_t r_gout = make_rsrc(gout, (size_t)g.B * g.O * g.S_o * 2);

  // W^T[tap] fragments of this wave's channel block: resident for the whole pixel range
  U4 wf[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks)
    wf[ks] = active ? wpb[(((int64_t)tap * hd.cblks + cblk) * NKS + ks) * 64 + lane] : U4{0, 0, 0, 0};

  f32x16 acc2[MB2];
#pragma unroll
  for (int i = 0; i < MB2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[i][r] = 0.f;

  // ---- grad_out tile: item = (o, pixel octet); two items per thread in flight, tail loop.
  // (gb, gp) = image / pixel of the first pixel of the NEXT tile to load (wave-uniform) ----
  const bool vec_ok = (g.S_o & 7) == 0, tile_ok = (g.S_o & 31) == 0;
  int gb = b_first, gp = p_first;
  auto load_item = [&](int item) -> U4 {
    const int o = item >> 2, oct = item & 3;
    int bb = gb, pp = gp + oct * 8;
    while (pp >= g.S_o) { pp -= g.S_o; ++bb; }
    U4 v = {0, 0, 0, 0};
    // 32 | S_o: a tile never straddles two images -- row offset per thread, tile offset scalar, and
    // images beyond the batch fall out of the buffer's range (no 64-bit address arithmetic per tile)
    if (tile_ok) return buf_load4u(r_gout, o < g.O ? (o * g.S_o + oct * 8) * 2 : kHpOob, (gb * g.O * g.S_o + gp) * 2);
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
  auto store_item = [&](int item, const U4 &v, int buf) {
    const int o = item >> 2, oct = item & 3;
    *reinterpret_cast<U4 *>(Gop + (buf * OpL + o) * kPP + oct * 8) = v;
  };
  const int ngitems = OpL * 4;
  U4 gi0 = {0, 0, 0, 0}, gi1 = {0, 0, 0, 0};
  int sb = b_first, sp0 = p_first;   // position of the tile whose tail items g_store loads itself
  int cb = b_first, cpx = p_first;   // image / pixel of the first pixel of the tile being processed
  auto g_load = [&]() {   // requests the tile at (gb, gp)
    if (tid < ngitems) gi0 = load_item(tid);
    if (tid + NTW < ngitems) gi1 = load_item(tid + NTW);
  };
  auto g_advance = [&]() {
    gp += 32;
    while (gp >= g.S_o) { gp -= g.S_o; ++gb; }
  };
  auto g_store = [&](int buf) {   // the tile requested by the previous g_load
    if (tid < ngitems) store_item(tid, gi0, buf);
    if (tid + NTW < ngitems) store_item(tid + NTW, gi1, buf);
    if (ngitems > 2 * NTW) {
      const int kb = gb, kp = gp;
      gb = sb; gp = sp0;
      for (int item = tid + 2 * NTW; item < ngitems; item += NTW) store_item(item, load_item(item), buf);
      gb = kb; gp = kp;
    }
    sp0 += 32;
    while (sp0 >= g.S_o) { sp0 -= g.S_o; ++sb; }
  };

  // ---- prologue: grad_out tiles of the first two tiles ----
  g_load();
  g_advance();
  g_store(0);
  if (t_lo + 1 < t_hi) { g_load(); g_advance(); }
  __syncthreads();   // prologue barrier (state of the first tile is in the table)

  for (int tile = t_lo; tile < t_hi; ++tile) {
    const int buf = (tile - t_lo) & 1;
    // ================= P2: GEMM-1 -> Gc =================
    if (active) {
      f32x16 gc;
#pragma unroll
      for (int r = 0; r < 16; ++r) gc[r] = 0.f;
      // B fragment (K = o, N = pixel) from the [o][pixel] tile: two transposing reads per k-step;
      // lane i of a 16-lane group addresses row (i >> 2), pixel quad (i & 3) of its 4 x 16 block
      const Raw *bp = Gop + (buf * OpL + o_base + 8 * kh + ((lane & 15) >> 2)) * kPP + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        U4 bfrag;
        lds_tr2(bp + ks * 16 * kPP, 4 * kPP, bfrag);
        gc = T::mfma(wf[ks], bfrag, gc);
      }
      float g0[8], g1[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) { g0[r] = gc[r]; g1[r] = gc[8 + r]; }
      Raw *dst = Gc + pl * pitch_gc + cblk * 32 + 16 * kh;
      *reinterpret_cast<U4 *>(dst) = pack8<T>(g0);
      *reinterpret_cast<U4 *>(dst + 8) = pack8<T>(g1);
    }
    __syncthreads();   // B2
    // ================= P3: gather role =================
    {
      // 2-D: both items' gathers are requested up front; 3-D (8 corners): one item at a time
      constexpr int NI = ND == 2 ? 2 : 1;
      U4 x[NI][NC], gq[NI];
      float wm[NI][NC];
      int grow[NI], gimg[NI];
      const int *st_tile = St + buf * 32 * g.DG * SW;
      const rsrc_t r_gcol = make_rsrc(gcol + (size_t)cb * gcol_img, gcol_img * 2);   // image of this tile (tile_ok)
      auto request = [&](int k, int slot) {
        if (it_on[k]) {
          const int *sp = st_tile + (it_p[k] * g.DG + it_dg[k]) * SW;
          int ev[SW];
#pragma unroll
          for (int q = 0; q < SW; q += 4) {
            const int4 e = *reinterpret_cast<const int4 *>(sp + q);
            ev[q] = e.x; ev[q + 1] = e.y; ev[q + 2] = e.z; ev[q + 3] = e.w;
          }
#pragma unroll
          for (int ci = 0; ci < NC; ++ci) {
            x[slot][ci] = buf_load4u(r_xt, ev[ci] + it_oc[k] * 16, 0);
            wm[slot][ci] = __int_as_float(ev[NC + ci]);
          }
          grow[slot] = ev[2 * NC];
          gimg[slot] = ev[2 * NC + 1];
          gq[slot] = *reinterpret_cast<const U4 *>(Gc + it_p[k] * pitch_gc + it_oc[k] * 8);
        }
      };
      auto consume = [&](int k, int slot) {
        if (it_on[k]) {
          if (tile_ok)   // all pixels of the tile in one image: scalar base, dead pixels out of range (dropped)
            buf_store4u_nt(r_gcol, grow[slot] + it_oc[k] * 16, 0, gq[slot]);
          else if (grow[slot] != kHpOob)
            *reinterpret_cast<U4 *>(gcol + (size_t)gimg[slot] * gcol_img + (grow[slot] >> 1) + it_oc[k] * 8) = gq[slot];
          float col[8], S[NC];
#pragma unroll
          for (int j = 0; j < 8; ++j) col[j] = 0.f;
#pragma unroll
          for (int ci = 0; ci < NC; ++ci) {
            S[ci] = dot8<T>(0.f, x[slot][ci], gq[slot]);
            mac8<T>(col, x[slot][ci], wm[slot][ci]);
          }
          *reinterpret_cast<U4 *>(Col + it_p[k] * pitch_gc + it_oc[k] * 8) = pack8<T>(col);
          // reduce S over the `sub` lanes that share (pixel, dg); partials -> LDS.  Up to 16 lanes
          // (one DPP row) with DPP operands, ds_bpermute only beyond
          hp_dpp_sum<NC>(S, sub);
          for (int d = 16; d < sub; d <<= 1)
#pragma unroll
            for (int ci = 0; ci < NC; ++ci) S[ci] += __shfl_xor(S[ci], d, 64);
          const int ol = it_oc[k] - it_dg[k] * LPD;
          if ((ol & (sub - 1)) == 0) {
            float *sp = Spart + (((it_p[k] * g.DG + it_dg[k]) * msub) + ol / sub) * NC;
#pragma unroll
            for (int ci = 0; ci < NC; ++ci) sp[ci] = S[ci];
          }
        }
      };
      request(0, 0);
      if (NI == 2) request(1, NI - 1);
      // the next tile's grad_out goes to the other LDS buffer while the gathers are in flight
      if (tile + 1 < t_hi) g_store(buf ^ 1);
      consume(0, 0);
      if (NI == 1) request(1, 0);
      consume(1, NI - 1);
    }
    __syncthreads();   // B3
    // ================= P4: GEMM-2 =================
    if (tile + 2 < t_hi) { g_load(); g_advance(); }
    if (active) {
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        // B fragment (K = pixel, N = channel) from the [pixel][c] column tile
        U4 bc;
        lds_tr2(Col + (ks2 * 16 + 8 * kh + ((lane & 15) >> 2)) * pitch_gc + cblk * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3),
                4 * pitch_gc, bc);
#pragma unroll
        for (int ob = 0; ob < MB2; ++ob) {
          const U4 a = *reinterpret_cast<const U4 *>(Gop + (buf * OpL + o_base + ob * 32 + pl) * kPP + ks2 * 16 + 8 * kh);
          acc2[ob] = T::mfma(a, bc, acc2[ob]);
        }
      }
    }
    cpx += 32;
    while (cpx >= g.S_o) { cpx -= g.S_o; ++cb; }
  }
  if (active) {
    float4 *dst = reinterpret_cast<float4 *>(
        part + ((((int64_t)tap * hd.ranges + range) * hd.cblks + cblk) * MB2) * 1024 + lane * 16);
#pragma unroll
    for (int ob = 0; ob < MB2; ++ob)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        dst[ob * 256 + q] = make_float4(acc2[ob][4 * q], acc2[ob][4 * q + 1], acc2[ob][4 * q + 2], acc2[ob][4 * q + 3]);
  }
}

}  // namespace


size_t hp_bwd2_lds_bytes(const Geom &g, const HpDims &hd) {
  const int nc = 1 << g.nd;
  const int lpp = hd.Cp / 8, lpd = g.DG == 1 ? lpp : g.Cdg / 8;
  int sub = 1;
  while (sub < 64 && lpd % (sub * 2) == 0) sub *= 2;
  const int msub = lpd / sub;
  return (size_t)2 * hd.OpL * kPP * 2 + (size_t)2 * 32 * (hd.Cp + 8) * 2 +
         (size_t)2 * 32 * g.DG * (2 * nc + 4) * 4 + (size_t)32 * g.DG * msub * nc * 4;
}

template <int ND, bool MOD, typename T, int WAVES, int NKS, int NS>
static int launch_bwd2_hp(const Geom &g, const HpDims &hd, const Tensors &t, const void *xt,
                          const void *wpb, const int4 *btab, void *gcol, float *part, int *cnt,
                          hipStream_t stream) {
  using Raw = typename T::Raw;
  const size_t lds = hp_bwd2_lds_bytes(g, hd);
  if (lds > 64 * 1024) {
    hipError_t ea = hipFuncSetAttribute((const void *)hp_bwd2_kernel<ND, MOD, T, WAVES, NKS, NS>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ea != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(ea)); return MDCONV_ELAUNCH; }
  }
  hp_debug_plan("hp_bwd2", hp_bwd2_kernel<ND, MOD, T, WAVES, NKS, NS>, 64 * (WAVES + NS), lds, (long)hd.ranges * g.K);
  hipLaunchKernelGGL((hp_bwd2_kernel<ND, MOD, T, WAVES, NKS, NS>), dim3(hd.ranges, g.K), dim3(64 * (WAVES + NS)), lds,
                     stream, g, hd, (const Raw *)xt, (const U4 *)wpb, btab, (const Raw *)t.grad_output,
                     (const Raw *)t.offset, (const Raw *)t.mask, (Raw *)gcol, (Raw *)t.grad_offset,
                     (Raw *)t.grad_mask, part, cnt);
  return check_launch("hp_bwd2");
}

template <int ND, bool MOD, typename T>
static int dispatch_bwd2_hp(const Geom &g, const HpDims &hd, const Tensors &t, const void *xt,
                            const void *wpb, const int4 *btab, void *gcol, float *part, int *cnt,
                            hipStream_t stream) {
#define HP_BWD(W, N)                                                                           \
  do {                                                                                         \
    if (g.DG > 2) return launch_bwd2_hp<ND, MOD, T, W, N, 2>(g, hd, t, xt, wpb, btab, gcol, part, cnt, stream); \
    return launch_bwd2_hp<ND, MOD, T, W, N, 1>(g, hd, t, xt, wpb, btab, gcol, part, cnt, stream);              \
  } while (0)
#define HP_BWD_W(W)                                                                            \
  switch (hd.nks) {                                                                            \
    case 2: HP_BWD(W, 2);                                                                      \
    case 4: HP_BWD(W, 4);                                                                      \
    case 8: HP_BWD(W, 8);                                                                      \
    default: HP_BWD(W, 16);                                                                    \
  }
  switch (hd.waves) {
    case 1: HP_BWD_W(1);
    case 2: HP_BWD_W(2);
    case 4: HP_BWD_W(4);
    default: HP_BWD_W(8);
  }
#undef HP_BWD_W
#undef HP_BWD
}

int hp_backward2_launch(const Geom &g, const HpDims &hd, int dtype, const Tensors &t, const void *xt,
                        const void *wpb, const int4 *btab, void *gcol, float *part, int *cnt,
                        hipStream_t stream) {
#define HP_DISPATCH(T)                                                                            \
  do {                                                                                            \
    if (g.nd == 2)                                                                                \
      return g.modulated ? dispatch_bwd2_hp<2, true, T>(g, hd, t, xt, wpb, btab, gcol, part, cnt, stream)  \
                         : dispatch_bwd2_hp<2, false, T>(g, hd, t, xt, wpb, btab, gcol, part, cnt, stream); \
    return g.modulated ? dispatch_bwd2_hp<3, true, T>(g, hd, t, xt, wpb, btab, gcol, part, cnt, stream)    \
                       : dispatch_bwd2_hp<3, false, T>(g, hd, t, xt, wpb, btab, gcol, part, cnt, stream);   \
  } while (0)
  if (dtype == MDCONV_F16) HP_DISPATCH(F16);
  HP_DISPATCH(BF16);
#undef HP_DISPATCH
}

}  // namespace mdconv
