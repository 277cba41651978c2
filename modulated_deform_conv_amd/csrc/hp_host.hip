// hp_host.hip -- dispatch, workspace layout and kernel sequence of the native 16-bit path.
//
// forward : pack weights -> channels-last input copy -> hp_fwd_kernel
// backward: pack W^T -> channels-last input copy -> hp_bwd3_kernel (GEMM-1 + coordinate gradients +
//           grad_col rows + column rows with ONE gather pass, CSR counting) -> hp_gemm2_kernel (dense
//           GEMM-2 over the column rows)   [shapes outside hp_bwd3: hp_bwd2_kernel, GEMM-2 fused in]
//           -> split-K reduce of grad_weight, grad_bias -> [weights-ready event] -> CSR scan + fill ->
//           col2im gather
// Calls whose channels-last copy would exceed 2 GiB (32-bit buffer offsets) are cut into batch
// chunks; grad_weight accumulates across chunks.
#include "hp_kernels.hpp"

#include <stdlib.h>

#include "mfma_kernels.hpp"

namespace mdconv {

int num_cus();   // mfma_bwd_data.hip

namespace {

size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

int pow2_ceil(int x) {
  int p = 1;
  while (p < x) p <<= 1;
  return p;
}

bool hp_enabled() {
  static const int on = getenv("MDCONV_HP") ? atoi(getenv("MDCONV_HP")) : 1;
  return on != 0;
}

size_t hp_chunk_limit() {
  static size_t lim = 0;
  if (!lim) {
    lim = (size_t)0x7e000000;   // below kHpOob
    const char *e = getenv("MDCONV_CHUNK_LIMIT_BYTES");
    if (e && atoll(e) > 0 && (size_t)atoll(e) < lim) lim = (size_t)atoll(e);
  }
  return lim;
}

Geom chunk_geom(const Geom &g, int bc) {
  Geom c = g;
  c.B = bc;
  c.N = bc * g.S_o;
  return c;
}

// which backward kernel: MDCONV_HP_BWD = 1 -> hp_bwd (lane = pixel), 2 -> hp_bwd2 (fused, tap-stationary),
// 3 (default) -> hp_bwd3 + hp_gemm2 where the shape qualifies and is large enough to fill the chip (use_bwd3), else as 2;
// 4 -> hp_bwd3 wherever it is supported (the test suite's way to reach every instance with small shapes)
int bwd_version() {
  static const int v = getenv("MDCONV_HP_BWD") ? atoi(getenv("MDCONV_HP_BWD")) : 3;
  return v;
}
// hp_bwd2 instances that hold W^T[tap] and the grad_W[tap] accumulators without spilling (tools/kres.py hp_bwd2: 16 k-steps,
// or 8 with the 8 + 2-wave workgroup, go to scratch) and shapes it takes at all (deformable groups of whole 32-channel blocks)
static bool bwd2_takes(const Geom &g, const HpDims &hd) {
  if (g.DG > 4 || (g.DG > 1 && g.Cdg % 32)) return false;
  return hp_bwd2_lds_bytes(g, hd) <= 160 * 1024;
}
static bool bwd2_spills(const HpDims &hd) { return hd.nks >= 16 || (hd.waves >= 8 && hd.nks >= 8); }
// Pixel-stationary (hp_bwd3 + hp_gemm2) or tap-stationary (hp_bwd2)?  hp_bwd3's workgroup walks ALL taps of its 128 pixels -- a
// grid of at most one workgroup per CU is one long latency chain per CU -- while hp_bwd2's grid is (tap, pixel range): parallel
// over the taps.  Measured crossover (profiles/r06_experiments.md 9: 27 shapes, both kernels): up to ~one 128-pixel tile per CU
// hp_bwd2 wins by 5-40 % (3-D 128 -> 128 at 4 x 14 x 14, B = 4: 0.56 -> 0.37 ms per step), beyond ~1.5 per CU hp_bwd3 does.
// The spilling instances of hp_bwd2 cross over earlier, at a tile count that falls with the channel count (hp_bwd2's time per
// (tile, tap) pair grows with C_in x C_out, hp_bwd3's latency chain per tap hardly) and is about twice as high in 3-D: measured
// on 21 shapes with 16 k-steps (profiles/r06_experiments.md 16) at 13-25 tiles for 2-D 256 -> 256, 25-49 for 3-D 256 -> 256,
// 25-98 for 2-D 128 -> 256, beyond 49 for 3-D 128 -> 256 -- tiles <= CUs x 16 (36 in 3-D) / C_in.  (Round 6 first shipped
// "(tile, tap) pairs <= CUs" from one data point: 3-D 256 -> 256 at 4 x 14 x 14, B = 2 went to hp_bwd3 at 0.88 ms against 0.56.)
bool use_bwd3(const Geom &g, const HpDims &hd) {
  if (bwd_version() < 3 || !hp_bwd3_supported(g, hd)) return false;
  if (bwd_version() == 3 && bwd2_takes(g, hd)) {
    const long tiles = (g.N + 127) / 128;
    const long limit = bwd2_spills(hd) ? (long)num_cus() * (g.nd == 3 ? 36 : 16) / hd.Cp : num_cus();
    if (tiles <= limit) return false;
  }
  return true;
}

struct FwdLayout { size_t off_xt, off_w, off_tab, total; };
struct BwdLayout { size_t off_xt, off_w, off_tab, off_gcol, off_col, off_part, off_gw32, off_cnt, off_rowptr, off_entries, off_sums, total; };

// grad_input gather: MDCONV_HP_C2I = 1 -> one pass (every row read 2^(nd-1) times), 2 (default) -> two passes
bool use_col2im2() {
  static const int v = getenv("MDCONV_HP_C2I") ? atoi(getenv("MDCONV_HP_C2I")) : 2;
  return v >= 2;
}

// images per chunk: channels-last input copy (and one image's grad_col) below the limit
int chunk_batch(const Geom &g, const HpDims &hd, bool backward) {
  const size_t lim = hp_chunk_limit();
  size_t per = (size_t)g.S_i * hd.Cp * 2;
  const size_t per_out = (size_t)g.S_o * (backward ? hd.Op : g.O) * 2;
  if (per_out > per) per = per_out;
  if (per >= lim) return 0;
  if (backward && (size_t)g.K * g.S_o * hd.Cp * 2 >= lim) return 0;   // one image's grad_col rows
  int bc = (int)(lim / per);
  return bc > g.B ? g.B : bc;
}

FwdLayout fwd_layout(const Geom &gc, const HpDims &hd) {
  FwdLayout L;
  size_t off = 0;
  L.off_xt = off;  off += align_up((size_t)gc.B * gc.S_i * hd.Cp * 2);
  L.off_w = off;   off += align_up((size_t)gc.K * (hd.Cp / 16) * hd.oblks * 1024);
  L.off_tab = off; off += align_up((size_t)hd.oranges * (hd.Cp / 16 + 1) * sizeof(int2));
  L.total = off;
  return L;
}

BwdLayout bwd_layout(const Geom &gc, const HpDims &hd, int dtype) {
  BwdLayout L;
  size_t off = 0;
  L.off_xt = off;   off += align_up((size_t)gc.B * gc.S_i * hd.Cp * 2);
  L.off_w = off;    off += align_up((size_t)gc.K * hd.cblks * hd.nks * 1024);
  L.off_tab = off;  off += align_up((size_t)hd.cblks * sizeof(int4));
  L.off_gcol = off; off += align_up((size_t)gc.B * gc.K * gc.S_o * hd.Cp * 2);
  L.off_col = off;  off += use_bwd3(gc, hd) ? align_up((size_t)gc.B * gc.K * gc.S_o * hd.Cp * 2) : 0;   // column rows for GEMM-2
  // a shorter last chunk can have MORE ranges than a full one (ranges is not monotonic in the tile
  // count), so the partials are sized for the bound; gw32 = running fp32 grad_weight over chunks
  L.off_part = off; off += align_up((size_t)gc.K * hd.max_ranges * hd.cblks * hd.MB2 * 4096);
  L.off_gw32 = off; off += align_up((size_t)gc.O * gc.Cg * gc.K * sizeof(float));
  // scatter lists: one 32-byte entry per sample, keyed by its extended anchor (hp_col2im.hip)
  const size_t S_e = (size_t)hp_anchor_space(gc);
  L.off_cnt = off;  off += align_up((size_t)gc.B * gc.DG * S_e * sizeof(int));
  L.off_rowptr = off; off += align_up((size_t)gc.B * gc.DG * (S_e + 1) * sizeof(int));
  L.off_entries = off; off += align_up((size_t)gc.B * gc.DG * gc.K * gc.S_o * 32);
  L.off_sums = off; off += use_col2im2() ? align_up(hp_col2im_sums_bytes(gc, hd, dtype)) : 0;
  L.total = off;
  return L;
}

}  // namespace


// workgroups per CU the fused backward kernel is sized for when it has 8 + 2 waves (C_in > 128)
static int hp_wg_per_cu8() {
  return 2;
}

HpDims hp_dims(const Geom &g) {
  HpDims hd;
  hd.Cp = (g.C + 31) / 32 * 32;
  hd.cblks = hd.Cp / 32;
  hd.Op = (g.O + 31) / 32 * 32;
  hd.oblks = hd.Op / 32;
  // forward: output-channel blocks per workgroup row.  At most 4 (64 accumulator registers): the 8-block instance
  // (128, one workgroup per CU) lost to two rows of 4 at EVERY batch size -- MDCN2d 256 -> 256 at 56 x 56 fp16: 477 -> 398 us at
  // B = 32, 146 -> 123 at B = 8 -- although the rows gather the same corners twice (L2 hits).  Grids of fewer than half a
  // workgroup per CU go down to single blocks: the kernel is one latency chain per workgroup there (B = 2: 96 -> 80 us,
  // 14 x 14 at B = 16: 89 -> 65 us; profiles/r06_experiments.md 8).
  hd.MB = hd.oblks >= 3 ? 4 : hd.oblks;
  if (g.G == 1 && (long)((g.N + 127) / 128) * ((hd.oblks + hd.MB - 1) / hd.MB) * 2 < num_cus()) hd.MB = 1;
  if (g.G > 1) {
    // conv groups: a workgroup row only needs the output channels one 64-channel K stage can
    // reach (cfg3: 64 channels = 8 groups = 64 outputs), so rows are made that narrow -- more,
    // lighter workgroups (fewer accumulators, one K stage per tap) instead of one row that walks
    // every input channel with 7 of its 8 output blocks idle
    const int reach = (64 / g.Cg < 1 ? 1 : 64 / g.Cg) * g.Og;   // outputs fed by 64 input channels
    int mb = pow2_ceil((reach + 31) / 32);
    if (mb < hd.MB) hd.MB = mb;
  }
  hd.oranges = (hd.oblks + hd.MB - 1) / hd.MB;
  hd.fwd_nmax = 1;
  for (int orange = 0; orange < hd.oranges; ++orange) {
    const int b_lo = orange * hd.MB, b_hi = (b_lo + hd.MB < hd.oblks ? b_lo + hd.MB : hd.oblks) - 1;
    for (int ch = 0; ch * 16 < g.C; ++ch) {
      const int g_lo = (ch * 16) / g.Cg, g_hi = (ch * 16 + 15 < g.C ? ch * 16 + 15 : g.C - 1) / g.Cg;
      const int ob_lo = (g_lo * g.Og) / 32 > b_lo ? (g_lo * g.Og) / 32 : b_lo;
      const int ob_hi = ((g_hi + 1) * g.Og - 1) / 32 < b_hi ? ((g_hi + 1) * g.Og - 1) / 32 : b_hi;
      if (ob_hi - ob_lo + 1 > hd.fwd_nmax) hd.fwd_nmax = ob_hi - ob_lo + 1;
    }
  }
  // backward: widest output-channel range (32-aligned) any 32-channel block needs
  int span = 32, base_max = 0;
  for (int cblk = 0; cblk < hd.cblks; ++cblk) {
    const int c_lo = cblk * 32 < g.C ? cblk * 32 : g.C - 1;
    const int c_hi = cblk * 32 + 31 < g.C ? cblk * 32 + 31 : g.C - 1;
    const int o_lo = (c_lo / g.Cg) * g.Og, o_hi = (c_hi / g.Cg + 1) * g.Og;
    const int base = o_lo / 32 * 32;
    const int s = (o_hi - base + 31) / 32 * 32;
    if (s > span) span = s;
    if (base > base_max) base_max = base;
  }
  hd.MB2 = pow2_ceil(span / 32);
  hd.nks = hd.MB2 * 2;
  hd.OpL = base_max + 32 * hd.MB2 > hd.Op ? base_max + 32 * hd.MB2 : hd.Op;
  hd.waves = pow2_ceil(hd.cblks);
  hd.ntiles = (g.N + 31) / 32;
  {
    static const int blocked_env = getenv("MDCONV_HP_BLOCKED") ? atoi(getenv("MDCONV_HP_BLOCKED")) : 1;
    hd.blocked = blocked_env && hp_blocked_ok(g) ? 1 : 0;
  }
  // pixel ranges per tap of the fused kernel: about one dispatch round of workgroups (2 workgroups of
  // 4 + 1 waves per CU, 1-2 of 8 + 2)
  const int slots = num_cus() * (hd.waves >= 8 ? hp_wg_per_cu8() : 2 * (4 / hd.waves));
  int ranges = slots / g.K;
  if (ranges < 1) ranges = 1;
  hd.max_ranges = ranges;
  if (ranges > hd.ntiles) ranges = hd.ntiles;
  hd.tiles_per_range = (hd.ntiles + ranges - 1) / ranges;
  hd.ranges = (hd.ntiles + hd.tiles_per_range - 1) / hd.tiles_per_range;
  // GEMM-2 (dense, HBM-bound): 4 workgroups per CU in flight, at least 8 tiles per workgroup.  With 8 output blocks per
  // wave (MB2 = 8: 128 accumulator registers, one resident workgroup per CU) one workgroup per CU: every range costs
  // K x cblks x MB2 x 4 KB of fp32 partials written and read back (256 -> 256 channels at 56 x 56, B = 8: 98 ranges = 231 MB
  // of partials beside 115 MB of column rows -- hp_gemm2 169 us + the reduction 50 us; 28 ranges: profiles/r06_experiments.md 2)
  int rw = num_cus() * (hd.MB2 > 4 ? 1 : 4) / g.K;
  // (one dispatch round -- 3 resident per CU -- or 2 per CU: the cfg5 backward moves by +-0.03 ms, profiles/r05_experiments.md 8)
  if (rw < 1) rw = 1;
  if (rw > hd.max_ranges) hd.max_ranges = rw;
  if (rw > (hd.ntiles + 7) / 8) rw = (hd.ntiles + 7) / 8;
  hd.tiles_per_range_w = (hd.ntiles + rw - 1) / rw;
  hd.ranges_w = (hd.ntiles + hd.tiles_per_range_w - 1) / hd.tiles_per_range_w;
  return hd;
}

// Deformable groups the native kernels cannot tile -- 24 / 48 / 80 ... channels: not whole pairs of 8-channel lanes, or not the
// power-of-two lane count the pixel-stationary backward gives a group -- run GROUP-PADDED (round 6): the kernels see groups of
// 32 / 64 / 128 channels (Geom::cm_pad), the four layout passes that touch the caller's C-indexed tensors (channels-last input
// copy, weight packing, grad_weight reduction, grad_input stencil) map the channels (caller_channel, mdconv_common.hpp), and
// the padding channels carry zero input and zero weights.  Up to 2x the channel work instead of DG single-group fp32 slices
// through workspace copies (fp16 96 -> 96 at 40 x 40, 4 groups: 1.11 -> 0.33 ms; profiles/r06_experiments.md 17).  Not for a
// channels-last input (read in place, C wide) nor with the one-pass gather (MDCONV_HP_C2I=1 writes grad_input itself).
static bool group_padded(const Geom &g, Geom *gv) {
  if (g.cm_pad || g.G != 1 || g.DG == 1 || g.in_cl || !use_col2im2()) return false;
  int cdp = (g.DG == 2 || g.DG == 4) ? pow2_ceil(g.Cdg) : (g.Cdg + 31) / 32 * 32;
  if (cdp < 16) cdp = 16;
  if (cdp == g.Cdg) return false;
  if (cdp > 8 * g.Cdg) return false;   // single-channel groups: 16x the gather work loses to the shape-generic kernels (0.74 vs 0.56 ms)
  *gv = g;
  gv->C = gv->Cg = g.DG * cdp;
  gv->Cdg = cdp;
  gv->cm_pad = cdp;
  gv->cm_real = g.Cdg;
  gv->C_caller = g.C;
  return true;
}

static bool hp_supported_as(const Geom &g, int dtype, bool backward);
// the geometry the native kernels run for `g`: g itself or its group-padded form
// One deformable group and 96 / 160 / 192 / 224 padded channels: the pixel-stationary backward gives a pixel a power-of-two lane count,
// so these widths ran on the tap-stationary kernels at any size; padded to 128 / 256 channels (the same channel map, one "group") they
// take hp_bwd3 where its size rule applies (fp16 192 -> 192 at 56 x 56, B = 8: 0.91 -> 0.42 ms; 224 -> 256: 1.00 -> 0.44; 3-D 160 -> 160
// at 8 x 28 x 28: 1.41 -> 1.05; small grids keep the tap-stationary kernels on the unpadded width; experiment log 25)
static bool width_padded(const Geom &g, Geom *gv) {
  if (g.cm_pad || g.G != 1 || g.DG != 1 || g.in_cl || !use_col2im2()) return false;
  const int Cp = (g.C + 31) / 32 * 32;
  if (Cp <= 64 || Cp >= 256 || pow2_ceil(Cp) == Cp) return false;
  *gv = g;
  gv->C = gv->Cg = gv->Cdg = pow2_ceil(Cp);
  gv->cm_pad = gv->C;
  gv->cm_real = g.C;
  gv->C_caller = g.C;
  return true;
}
static bool hp_plan_geom(const Geom &g, int dtype, bool backward, Geom *ge) {
  if (hp_supported_as(g, dtype, backward)) {
    if (backward && width_padded(g, ge) && hp_supported_as(*ge, dtype, true)) {
      const int bc = chunk_batch(*ge, hp_dims(*ge), true);
      if (bc > 0) {
        const Geom gc = chunk_geom(*ge, bc);
        if (use_bwd3(gc, hp_dims(gc))) return true;
      }
    }
    *ge = g;
    return true;
  }
  return group_padded(g, ge) && hp_supported_as(*ge, dtype, backward);
}
bool hp_supported(const Geom &g, int dtype, bool backward) {
  Geom ge;
  return hp_plan_geom(g, dtype, backward, &ge);
}

static bool hp_supported_as(const Geom &g, int dtype, bool backward) {
  if (!hp_enabled()) return false;
  if (dtype != MDCONV_F16 && dtype != MDCONV_BF16) return false;
  if (g.DG > 1 && g.Cdg % 16) return false;
  const HpDims hd = hp_dims(g);
  // backward with deformable groups of 16 / 48 / ... channels: the pixel-stationary kernel only (its lanes own 8 channels
  // of one group each); the tap-stationary kernels reduce the coordinate sums per 32-channel block
  if (backward && g.DG > 1 && g.Cdg % 32 && !use_bwd3(g, hd)) return false;
  if (backward) {
    if (hd.cblks > 8 || hd.MB2 > 8) return false;   // one workgroup covers all input channels
    if (g.in_sz[g.nd - 1] < 2) return false;        // pair-keyed scatter lists
  }
  return chunk_batch(g, hd, backward) > 0;
}

// Forward of a FEW pixel tiles over MANY K stages: hp_fwd2 runs one workgroup per (128-pixel tile, output-channel row)
// through every tap and 64-channel stage, so a grid of a dozen workgroups takes one whole tile time on an otherwise empty
// chip (C = 512, 7 x 7, B = 16: 231 us; 3-D C = 256, 4 x 7 x 7, B = 4: 547 us) -- whereas the fp32 matrix forwards cut such
// grids into tap ranges (fwd_tail_plan) and take 87 / 109 us for the same shapes.  Such calls run on the fp32 kernels
// through fp32 copies (the route of every 16-bit shape the native kernels do not take: fp32 accumulation, one rounding of
// the output).  Not with a channels-last input (only the native kernels read it in place), not when MDCONV_HP_FWD selects a
// kernel explicitly (the forced-path tests).
bool hp_forward_preferred(const Geom &gcall, int dtype) {
  static const bool forced = getenv("MDCONV_HP_FWD") != nullptr;
  if (forced) return true;
  Geom g;
  if (!hp_plan_geom(gcall, dtype, false, &g)) return false;
  // counted in rows of 8 output blocks whatever rows hp_dims picks: single-block rows (small grids, MB = 1) multiply the
  // workgroups, not the work one of them finishes per unit time (2048 -> 512 at 7 x 7, B = 8: 64 single-block workgroups
  // 675 us, the fp32 route 277 us; profiles/r06_experiments.md 15)
  const int oblks = (g.O + 31) / 32;
  const long wgs = (long)((g.N + 127) / 128) * ((oblks + 7) / 8);
  const long stages = (long)g.K * ((g.C + 63) / 64);
  if (wgs > 16 || stages < 64) return true;
  return !mfma_supported(gcall, dtype, false);
}

size_t hp_workspace_bytes(const Geom &gcall, int dtype, bool backward) {
  Geom g;
  if (!hp_plan_geom(gcall, dtype, backward, &g)) return 0;
  HpDims hd = hp_dims(g);
  const int bc = chunk_batch(g, hd, backward);
  if (bc <= 0) return 0;
  const Geom gc = chunk_geom(g, bc);
  hd = hp_dims(gc);
  return backward ? bwd_layout(gc, hd, dtype).total : fwd_layout(gc, hd).total;
}

// hp_fwd2 with deformable groups: every workgroup row's channel range must start on a group
// boundary (its stages are numbered from there)
static bool fwd2_rows_align(const Geom &g, const HpDims &hd) {
  if (hd.oranges == 1) return true;
  if (g.G == 1) return false;
  for (int r = 1; r < hd.oranges; ++r) {
    const int o_first = r * hd.MB * 32;
    if (o_first % g.Og) return false;                      // rows start on a conv-group boundary
    const int c_first = (o_first / g.Og) * g.Cg;
    if (c_first % g.Cdg || c_first % 64) return false;     // ... which is a deformable-group / stage boundary
  }
  return true;
}

int hp_forward(const Geom &gcall, int dtype, const Tensors &t, void *ws, hipStream_t stream) {
  Geom g;   // the caller's geometry or its group-padded form
  if (!hp_plan_geom(gcall, dtype, false, &g)) { set_error("hp_forward: no plan"); return MDCONV_EUNSUPPORTED; }
  const int Bc = chunk_batch(g, hp_dims(g), false);
  if (Bc <= 0) { set_error("hp_forward: no plan"); return MDCONV_EUNSUPPORTED; }
  const Geom g0 = chunk_geom(g, Bc);
  const HpDims hd0 = hp_dims(g0);
  const FwdLayout L = fwd_layout(g0, hd0);
  char *base = (char *)ws;
  const int nc_off = g.DG * g.nd * g.K, nc_m = g.DG * g.K;
  int rc;
  if ((rc = hp_pack_fwd_weights(g0, hd0, dtype, t.weight, base + L.off_w, (int2 *)(base + L.off_tab), stream)))
    return rc;
  for (int b0 = 0; b0 < g.B; b0 += Bc) {
    const int bc = g.B - b0 < Bc ? g.B - b0 : Bc;
    const Geom gc = chunk_geom(g, bc);
    const HpDims hd = hp_dims(gc);
    Tensors tc = t;
    tc.input = (const char *)t.input + (size_t)b0 * caller_channels(g) * g.S_i * 2;
    tc.offset = (const char *)t.offset + (size_t)b0 * nc_off * g.S_o * 2;
    tc.mask = t.mask ? (const char *)t.mask + (size_t)b0 * nc_m * g.S_o * 2 : nullptr;
    tc.output = (char *)t.output + (size_t)b0 * g.O * g.S_o * 2;
    const void *xt = base + L.off_xt;
    if (g.in_cl) xt = (const char *)t.input + (size_t)b0 * g.S_i * g.C * 2;   // already channels-last
    else if ((rc = hp_nchw_to_nhwc(gc, hd, tc.input, base + L.off_xt, stream))) return rc;
    // quad-contiguous gathers (hp_fwd2.hip) unless a 64-channel stage would straddle deformable groups
    static const int fwd_ver = getenv("MDCONV_HP_FWD") ? atoi(getenv("MDCONV_HP_FWD")) : 2;
    const bool fwd2 = fwd_ver == 2 && (g.DG == 1 || (g.Cdg % 64 == 0 && fwd2_rows_align(g, hd)));
    profile_mark(0, true, stream, fwd2 ? "hp_fwd2_kernel" : "hp_fwd_kernel");
    if (fwd2)
      rc = hp_forward2_launch(gc, hd, dtype, tc, xt, base + L.off_w,
                              (const int2 *)(base + L.off_tab), stream);
    else
      rc = hp_forward_launch(gc, hd, dtype, tc, xt, base + L.off_w,
                             (const int2 *)(base + L.off_tab), stream);
    profile_mark(0, false, stream);
    if (rc) return rc;
  }
  return MDCONV_OK;
}

int hp_backward(const Geom &gcall, int dtype, const Tensors &t, void *ws, hipStream_t stream) {
  Geom g;   // the caller's geometry or its group-padded form
  if (!hp_plan_geom(gcall, dtype, true, &g)) { set_error("hp_backward: no plan"); return MDCONV_EUNSUPPORTED; }
  const int Bc = chunk_batch(g, hp_dims(g), true);
  if (Bc <= 0) { set_error("hp_backward: no plan"); return MDCONV_EUNSUPPORTED; }
  const Geom g0 = chunk_geom(g, Bc);
  const HpDims hd0 = hp_dims(g0);
  const BwdLayout L = bwd_layout(g0, hd0, dtype);
  char *base = (char *)ws;
  const int nc_off = g.DG * g.nd * g.K, nc_m = g.DG * g.K;
  int rc;
  if ((rc = hp_pack_bwd_weights(g0, hd0, dtype, t.weight, base + L.off_w, (int4 *)(base + L.off_tab), stream)))
    return rc;
  if (g.with_bias && (rc = hp_grad_bias(g, dtype, t.grad_output, t.grad_bias, stream))) return rc;
  for (int b0 = 0; b0 < g.B; b0 += Bc) {
    const int bc = g.B - b0 < Bc ? g.B - b0 : Bc;
    Geom gc = chunk_geom(g, bc);
    const bool multi = Bc < g.B, first = b0 == 0, last = b0 + bc >= g.B;
    const HpDims hd = hp_dims(gc);
    Tensors tc = t;
    tc.input = (const char *)t.input + (size_t)b0 * caller_channels(g) * g.S_i * 2;
    tc.offset = (const char *)t.offset + (size_t)b0 * nc_off * g.S_o * 2;
    tc.mask = t.mask ? (const char *)t.mask + (size_t)b0 * nc_m * g.S_o * 2 : nullptr;
    tc.grad_output = (const char *)t.grad_output + (size_t)b0 * g.O * g.S_o * 2;
    tc.grad_input = (char *)t.grad_input + (size_t)b0 * caller_channels(g) * g.S_i * 2;
    tc.grad_offset = (char *)t.grad_offset + (size_t)b0 * nc_off * g.S_o * 2;
    tc.grad_mask = t.grad_mask ? (char *)t.grad_mask + (size_t)b0 * nc_m * g.S_o * 2 : nullptr;
    int *cnt = (int *)(base + L.off_cnt), *rowptr = (int *)(base + L.off_rowptr);
    const void *xt = base + L.off_xt;
    if (g.in_cl) xt = (const char *)t.input + (size_t)b0 * g.S_i * g.C * 2;   // already channels-last
    else if ((rc = hp_nchw_to_nhwc(gc, hd, tc.input, base + L.off_xt, stream))) return rc;
    if ((rc = hp_csr_zero(gc, cnt, stream))) return rc;
    const bool bwd3 = use_bwd3(gc, hd);
    const bool bwd2 = !bwd3 && bwd_version() >= 2 && g.DG <= 4 && hp_bwd2_lds_bytes(gc, hd) <= 160 * 1024;
    profile_mark(1, true, stream, bwd3 ? "hp_bwd3_kernel" : (bwd2 ? "hp_bwd2_kernel" : "hp_bwd_kernel"));
    if (bwd3)
      rc = hp_backward3_launch(gc, hd, dtype, tc, xt, base + L.off_w, base + L.off_gcol, base + L.off_col, cnt, stream);
    else if (bwd2)
      rc = hp_backward2_launch(gc, hd, dtype, tc, xt, base + L.off_w,
                               (const int4 *)(base + L.off_tab), base + L.off_gcol,
                               (float *)(base + L.off_part), cnt, stream);
    else
      rc = hp_backward_launch(gc, hd, dtype, tc, xt, base + L.off_w,
                              (const int4 *)(base + L.off_tab), base + L.off_gcol,
                              (float *)(base + L.off_part), cnt, stream);
    profile_mark(1, false, stream);
    if (rc) return rc;
    // Two independent tails: GEMM-2 -> split-K reduce (needs the column rows / partials) and the grad_input
    // gather (CSR scan + fill -> partial sums -> stencil; needs the grad_col rows and the counters).  Each
    // alone streams at ~3.6 TB/s; forked (mfma_kernels.hpp) they share the chip: cfg5 backward 5.94 -> 5.85 ms
    // (GEMM-2 1.0 -> 1.5 ms beside the gather).  Only where GEMM-2 is its own kernel: with the fused backward
    // the weight tail is one 27 us reduction and the fork's two cross-stream waits cost as much (cfg3).
    hipStream_t gs = bwd3 ? fork_side_stream(stream) : nullptr;
    const bool forked = gs != nullptr;
    if (!forked) gs = stream;
    auto weight_tail = [&]() -> int {
      int r;
      if (bwd3) {
        profile_mark(2, true, stream, "hp_gemm2_kernel");
        r = hp_gemm2_launch(gc, hd, dtype, tc, (const int4 *)(base + L.off_tab), base + L.off_col,
                            (float *)(base + L.off_part), stream);
        profile_mark(2, false, stream);
        if (r) return r;
      }
      if ((r = hp_reduce_grad_weight(gc, hd, bwd3 ? hd.ranges_w : hd.ranges, dtype, (const float *)(base + L.off_part),
                                     (const int4 *)(base + L.off_tab), t.grad_weight,
                                     multi ? (float *)(base + L.off_gw32) : nullptr, first, last, stream)))
        return r;
      return last ? record_weight_ready(stream) : MDCONV_OK;
    };
    if (!forked && (rc = weight_tail())) return rc;
    rc = hp_csr_build(gc, dtype, tc, cnt, rowptr, base + L.off_entries, gs);
    if (!rc) {
      profile_mark(3, true, gs, use_col2im2() ? "hp_col2im_sums_kernel" : "hp_col2im_kernel");
      rc = use_col2im2() ? hp_col2im2(gc, hd, dtype, tc, base + L.off_gcol, rowptr, base + L.off_entries, base + L.off_sums, gs)
                         : hp_col2im(gc, hd, dtype, tc, base + L.off_gcol, rowptr, base + L.off_entries, gs);
      profile_mark(3, false, gs);
    }
    if (forked) {
      // join on EVERY path once the side stream has work: an unjoined fork would leave it reading and writing the
      // caller's workspace and grad_input after an error return, and an open stream capture invalid
      if (!rc) rc = weight_tail();
      const int rj = join_side_stream(stream);
      if (!rc) rc = rj;
    }
    if (rc) return rc;
  }
  return MDCONV_OK;
}

}  // namespace mdconv
