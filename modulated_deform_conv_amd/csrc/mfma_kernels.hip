// mfma_kernels.hip -- dispatch, workspace layout and weight packing for the MFMA path.
#include "mfma_kernels.hpp"
#include "mfma_tile.hpp"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

namespace mdconv {

namespace {

// benchmark hooks: process-wide, guarded so that a threaded host cannot corrupt the event lists
std::atomic<bool> g_prof_on{false};
std::mutex g_prof_mu;
struct ProfPair { hipEvent_t a, b; };
constexpr int kProfSlots = 5;   // forward GEMM, backward data GEMM, backward weight GEMM, grad_input gather, coordinate gradients
std::vector<ProfPair> g_prof[kProfSlots];
size_t g_prof_used[kProfSlots] = {0, 0, 0, 0, 0};
const char *g_prof_name[kProfSlots] = {"", "", "", "", ""};

// W[g*Og + o][c][tap]  ->  wp (MFMA-fragment order, mfma_tile.hpp) and wq[g][tap][o][c], zero padded.
__global__ __launch_bounds__(256) void pack_weights_kernel(Geom g, PackDims pd,
                                                           const float *__restrict__ w,
                                                           float *__restrict__ wp,
                                                           float *__restrict__ wq) {
  const int64_t total = (int64_t)g.G * g.K * pd.Cgp * pd.Ogp;
  const int mblks = pd.Ogp / 32, cchunks = pd.Cgp / kBK;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    // decode i as a wp index: [grp][tap][cchunk][mblk][q][lane][s]
    int64_t r = i;
    const int s = (int)(r & 3); r >>= 2;
    const int lane = (int)(r & 63); r >>= 6;
    const int q = (int)(r & 1); r >>= 1;
    const int mblk = (int)(r % mblks); r /= mblks;
    const int cchunk = (int)(r % cchunks); r /= cchunks;
    const int tap = (int)(r % g.K);
    const int grp = (int)(r / g.K);
    const int o = mblk * 32 + (lane & 31);
    const int c = cchunk * kBK + 8 * q + 4 * (lane >> 5) + s;
    const float v = (o < g.Og && c < g.Cg)
                        ? w[((int64_t)(grp * g.Og + o) * g.Cg + c) * g.K + tap] : 0.f;
    wp[i] = v;
    if (wq) wq[(((int64_t)grp * g.K + tap) * pd.Ogp + o) * pd.Cgp + c] = v;
  }
}

size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

bool bwd_fork_enabled();   // defined with the fork / join helpers below

// scratch and gradients are cleared by kernels rather than hipMemsetAsync: memset nodes made HIP
// graph replay fault (tools/graph_check.py), and a plain kernel sequence captures cleanly
__global__ __launch_bounds__(256) void zero_words_kernel(unsigned *__restrict__ p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0u;
}
__global__ __launch_bounds__(256) void zero_halfwords_kernel(unsigned short *__restrict__ p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0;
}

}  // namespace

int device_cus() {
  static std::atomic<int> cus{0};
  int n = cus.load(std::memory_order_relaxed);
  if (n) return n;
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
    n = prop.multiProcessorCount;
  else
    n = 256;
  (void)hipGetLastError();
  cus.store(n, std::memory_order_relaxed);
  return n;
}

int zero_bytes(void *p, size_t bytes, hipStream_t s) {
  if (bytes == 0 || p == nullptr) return MDCONV_OK;
  const bool words = bytes % 4 == 0 && ((uintptr_t)p & 3) == 0;
  const int64_t n = (int64_t)(words ? bytes / 4 : bytes / 2);
  const int64_t blocks = (n + 255) / 256;
  const dim3 grid((unsigned)(blocks > 8192 ? 8192 : blocks));
  if (words) hipLaunchKernelGGL(zero_words_kernel, grid, dim3(256), 0, s, (unsigned *)p, n);
  else hipLaunchKernelGGL(zero_halfwords_kernel, grid, dim3(256), 0, s, (unsigned short *)p, n);
  return check_launch("zero");
}

void profile_mark(int which, bool begin, hipStream_t stream, const char *name) {
  if (!g_prof_on.load(std::memory_order_relaxed) || which < 0 || which >= kProfSlots) return;
  std::lock_guard<std::mutex> lock(g_prof_mu);
  if (name) g_prof_name[which] = name;
  if (begin) {
    if (g_prof_used[which] == g_prof[which].size()) {
      ProfPair p;
      if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return;
      g_prof[which].push_back(p);
    }
    (void)hipEventRecord(g_prof[which][g_prof_used[which]].a, stream);
  } else if (g_prof_used[which] < g_prof[which].size()) {
    (void)hipEventRecord(g_prof[which][g_prof_used[which]].b, stream);
    ++g_prof_used[which];
  }
}

int pack_weights_f32(const Geom &g, const PackDims &pd, const float *weight, float *wp, float *wq,
                     hipStream_t stream) {
  const int64_t total = (int64_t)g.G * g.K * pd.Cgp * pd.Ogp;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, stream, g, pd, weight, wp, wq);
  return check_launch("pack_weights");
}

BwdDims bwd_dims(const Geom &g) {
  BwdDims bd;
  bd.Np = (g.N + 31) / 32 * 32;
  // 64 x 64 tiles (all four waves busy when C_out <= 64) were measured SLOWER than 256 x 32 tiles
  // with idle waves at cfg4 for the NCHW gathers (GEMM-2 5.4 -> 8 ms: bound by the cache-line
  // traffic of the 8-corner gathers, not by the matrix work), so that variant stays an experiment.
  // With channels-last gathers the tile is always 64 input channels wide.
  bd.cl = bwd_channels_last(g) ? 1 : 0;
  bd.wtile = 0;
  if (bd.cl) bd.wtile = g.O <= 64 ? 1 : (g.O <= 128 ? 2 : 3);
  const int rm = bd.cl ? (bd.wtile == 1 ? 64 : (bd.wtile == 2 ? 128 : 256)) : (bd.wtile ? 64 : 256);
  const int cn = bd.cl ? 64 : (bd.wtile ? 64 : 32);
  bd.OgpB = (g.O + rm - 1) / rm * rm;
  bd.mblks = bd.OgpB / 32;
  bd.mtiles = bd.OgpB / rm;
  bd.Cp = (g.C + cn - 1) / cn * cn;
  bd.cblks = bd.Cp / cn;
  const int col_tiles = bd.mtiles * g.K * bd.cblks;
  const int pairs = bd.Np / 32;
  // Split-K so that the grid is ONE full dispatch round: every workgroup does the same work, so a grid of
  // 1.36 x the resident slots (cfg2 in round 3: 36 column tiles x 29 splits = 1044 workgroups on 256 CUs x 3)
  // runs its last 276 workgroups one per CU at half the matrix rate -- 0.68 of peak where the steady state
  // reaches 0.8+.  slots = CUs x resident workgroups of the instance that will run (hipOccupancy).
  const bool padn = bd.Np != g.N;
  const int occ = bd.cl ? mfma_bwd_weight_cl_occupancy(g.nd, padn, bd.wtile)
                        : mfma_bwd_weight_occupancy(g.nd, padn, bd.wtile);
  // ... minus one per CU when the grad_input gather runs beside this kernel on the forked stream: a full round of
  // 164-register workgroups leaves the gather no wave slot until the round retires, and the two tails then run one
  // after the other (cfg2, 36 column tiles: 21 splits = 756 workgroups -> GEMM-2 1.00 ms then gather 0.30 ms,
  // backward 2.35 ms; 14 splits = 504 workgroups, two per CU -> both done after 0.96 ms, backward 2.25 ms)
  const int occ_run = bwd_fork_enabled() && occ > 1 ? occ - 1 : occ;
  const int slots = device_cus() * occ_run;
  int splits = slots / col_tiles;
  if (splits > pairs) splits = pairs;
  if (splits < 1) splits = 1;
  bd.pairs_per_split = (pairs + splits - 1) / splits;
  bd.splits = (pairs + bd.pairs_per_split - 1) / bd.pairs_per_split;
  static const bool debug_plan = getenv("MDCONV_DEBUG_PLAN") != nullptr;
  if (debug_plan)
    fprintf(stderr, "[mdconv] GEMM-2 plan: cl %d wtile %d col_tiles %d occ %d slots %d splits %d x %d pairs\n", bd.cl,
            bd.wtile, col_tiles, occ, slots, bd.splits, bd.pairs_per_split);
  bd.ochunks = (g.O + 63) / 64 * 4;   // K loop of GEMM-1 is unrolled 4x
  bd.waves_c = g.C > 128 ? 4 : (g.C > 64 ? 2 : 1);
  // The grad_out tile ([32 * 4 / waves_c pixels] x C_out) lives in LDS: when it does not fit with the natural wave split
  // (C_in <= 64 and C_out > ~192 in 3-D: 128 pixels x 256 channels = 133 KB + the drain's tiles), more waves go along the
  // channels -- the extra ones own zero-padded channel blocks and idle through the drain -- and the pixel tile shrinks
  // with them.  Half the matrix rate of GEMM-1 at such shapes, against the shape-generic kernels they used to fall to (~10x).
  for (;;) {
    bd.cblks_q = (g.C + 64 * bd.waves_c - 1) / (64 * bd.waves_c) * (2 * bd.waves_c);
    // GEMM-1 drain: channels-last (line-wide gathers through an LDS hand-over, mfma_bwd_data.hip)
    // whenever the backward has the channels-last copy -- except for straight-line 2-D shapes whose
    // LDS (grad_out tile + parked accumulators) would then allow only one workgroup per CU
    // (C_in = 128, C_out = 256: 1.56 -> 1.70 ms).  MDCONV_BD_CL = 0 / 1 overrides.
    // Reduction buffer [tap group][64-channel blocks to reduce][nd + 1][pixels of the tile]: groups of
    // 3 taps instead of 9 where that keeps the kernel under 80 KB of LDS (two workgroups per CU).
    const int red_per_tap = 128 * (g.DG > 1 ? bd.cblks_q / (2 * bd.waves_c) : 1) * (g.nd + 1);
    const bool single_owner = g.DG == 1 && bd.waves_c == 1 && bd.cblks_q == 2;   // direct writes, no buffer
    auto size_red = [&]() {
      bd.tap_group = 9;
      bd.red_floats = single_owner ? 0 : bd.tap_group * red_per_tap;
      if (bd.red_floats && bwd_data_lds_bytes(g, bd) > 80 * 1024) {
        bd.tap_group = 3;
        bd.red_floats = bd.tap_group * red_per_tap;
      }
    };
    {
      const int nbatch = g.nd == 2 ? 4 : 8, nquads = bd.ochunks / 4;
      const bool straight = g.nd == 2 && g.G == 1 && nquads % nbatch == 0 && nquads / nbatch <= 2;
      static const int bd_cl_env = getenv("MDCONV_BD_CL") ? atoi(getenv("MDCONV_BD_CL")) : -1;
      bd.cl_drain = bd.cl;
      size_red();
      if (bd.cl && straight && (bd_cl_env == 0 || (bd_cl_env < 0 && bwd_data_lds_bytes(g, bd) > 80 * 1024))) {
        bd.cl_drain = 0;
        size_red();
      }
    }
    if (bwd_data_lds_bytes(g, bd) <= kBwdDataLdsCap || bd.waves_c == 4) break;
    bd.waves_c *= 2;
  }
  const int nc = 1 << g.nd;
  size_t off = 0;
  bd.off_wq = off;   off += align_up((size_t)g.K * bd.ochunks * bd.cblks_q * 2 * 64 * 16);
  bd.off_ga = off;   off += align_up((size_t)bd.Np * bd.OgpB * sizeof(float));
  bd.off_table = off; off += align_up((size_t)g.DG * g.K * bd.Np * 2 * (1 << g.nd) * sizeof(int));
  bd.off_part = off; off += align_up((size_t)bd.splits * g.K * bd.OgpB * bd.Cp * sizeof(float));
  bd.off_gcol = off; off += align_up((size_t)g.B * g.C * g.K * g.S_o * sizeof(float));
  // scatter lists: 2-D one entry per corner pair keyed by the pair's first pixel; 3-D one entry per
  // sample keyed by its low corner in the extended anchor space (4x fewer entries and atomics)
  bd.sample_keyed = g.nd == 3 ? 1 : 0;
  bd.S_e = bd.sample_keyed ? hp_anchor_space(g) : g.S_i;
  bd.off_cnt = off;  off += align_up((size_t)g.B * g.DG * bd.S_e * sizeof(int));
  bd.off_rowptr = off; off += align_up((size_t)g.B * g.DG * (bd.S_e + 1) * sizeof(int));
  bd.off_entries = off; off += align_up((size_t)g.B * g.DG * g.K * g.S_o * (bd.sample_keyed ? 32 : (nc / 2) * 16));
  bd.bias_tiles = (g.N + 32 * (4 / bd.waves_c) - 1) / (32 * (4 / bd.waves_c));
  bd.off_bias = off; off += align_up((size_t)bd.bias_tiles * g.O * sizeof(float));
  bd.off_xt = off;   off += bd.cl ? align_up((size_t)g.B * g.S_i * g.C * sizeof(float)) : 0;
  static const int c2i_env = getenv("MDCONV_C2I3D") ? atoi(getenv("MDCONV_C2I3D")) : 2;
  bd.two_pass = bd.sample_keyed && c2i_env >= 2 ? 1 : 0;
  bd.off_sums = off; off += bd.two_pass ? align_up(col2im3d_sums_bytes(g)) : 0;
  bd.off_bstage = off; off += g.with_bias ? align_up(grad_bias_stage_bytes(g)) : 0;
  bd.off_end = off;
  return bd;
}

// ---------------------------------------------------------------------------------------------
// Execution plan: batch chunking + fp16 I/O.
//  * The kernels address tensors with 32-bit byte offsets (raw buffer loads), so a call is cut
//    into chunks of Bc images such that every per-chunk tensor (and the grad_col workspace)
//    stays below 2 GiB.  grad_weight / grad_bias accumulate across chunks by construction.
//    (This is the only thing left of the reference's `in_step` chunk loop.)
//  * fp16 tensors are computed in fp32: each chunk is widened into fp32 copies in the workspace,
//    run through the fp32 kernels (coordinates and accumulation in fp32) and
//    narrowed back; grad_weight / grad_bias are accumulated in fp32 over all chunks.
// ---------------------------------------------------------------------------------------------
namespace {

// 2 GiB minus slack; MDCONV_CHUNK_LIMIT_BYTES lowers it so tests can force multi-chunk execution
size_t chunk_limit() {
  static size_t lim = 0;
  if (!lim) {
    lim = ((size_t)1 << 31) - (1 << 16);
    const char *e = getenv("MDCONV_CHUNK_LIMIT_BYTES");
    if (e && atoll(e) > 0 && (size_t)atoll(e) < lim) lim = (size_t)atoll(e);
  }
  return lim;
}

struct Plan {
  int Bc;
  bool half_io;
  Geom gc;            // geometry of a full chunk
  size_t core_bytes;  // workspace of the fp32 kernels for one chunk
  size_t off_w, off_b, off_x, off_off, off_m, off_go, off_out, off_gi, off_goff, off_gm, off_gw, off_gb;
  size_t total;
};

Geom chunk_geom(const Geom &g, int bc) {
  Geom c = g;
  c.B = bc;
  c.N = bc * g.S_o;
  return c;
}

size_t core_bytes_for(const Geom &gc, bool backward) {
  if (backward) return bwd_dims(gc).off_end;
  const PackDims pd = pack_dims(gc);
  size_t n = align_up((size_t)gc.G * gc.K * pd.Cgp * pd.Ogp * sizeof(float));
  if (fwd_channels_last(gc)) n += align_up(fwd_cl_bytes(gc));   // NHWC copy of the input chunk
  n += align_up(fwd_tail_bytes(gc));                            // tap-range partials of the last dispatch round
  return n;
}

bool make_plan(const Geom &g, int dtype, bool backward, Plan *p) {
  const size_t per_in = (size_t)g.C * g.S_i * 4, per_out = (size_t)g.O * g.S_o * 4;
  const size_t per_col = (size_t)g.C * g.K * g.S_o * 4;
  size_t per = per_in > per_out ? per_in : per_out;
  if (backward) {
    // every workspace buffer the backward kernels address with 32-bit buffer offsets must stay
    // below the limit for one chunk: grad_col, the packed grad_out (rows padded to the GEMM-2 tile:
    // up to 256 output channels even for small C_out), the tap table and the scatter lists
    const BwdDims b1 = bwd_dims(chunk_geom(g, 1));
    const size_t per_ga = (size_t)g.S_o * b1.OgpB * 4;
    const size_t per_tab = (size_t)g.DG * g.K * g.S_o * 2 * (1 << g.nd) * 4;
    const size_t per_ent = (size_t)g.DG * g.K * g.S_o * 32;   // 2 pair entries (2-D) or 1 sample entry (3-D)
    if (per_col > per) per = per_col;
    if (per_ga > per) per = per_ga;
    if (per_tab > per) per = per_tab;
    if (per_ent > per) per = per_ent;
  }
  const size_t kLim = chunk_limit();
  if (per >= kLim) return false;
  int bc = (int)(kLim / per);
  if (bc > g.B) bc = g.B;
  p->Bc = bc;
  p->half_io = dtype == MDCONV_F16 || dtype == MDCONV_BF16;
  p->gc = chunk_geom(g, bc);
  p->core_bytes = core_bytes_for(p->gc, backward);
  size_t off = p->core_bytes;
  auto take = [&](size_t &slot, size_t elems) { slot = off; off += align_up(elems * sizeof(float)); };
  p->off_w = p->off_b = p->off_x = p->off_off = p->off_m = p->off_go = p->off_out = 0;
  p->off_gi = p->off_goff = p->off_gm = p->off_gw = p->off_gb = 0;
  if (p->half_io) {
    const int nc_off = g.DG * g.nd * g.K, nc_m = g.DG * g.K;
    take(p->off_w, (size_t)g.O * g.Cg * g.K);
    take(p->off_b, (size_t)g.O);
    take(p->off_x, (size_t)bc * g.C * g.S_i);
    take(p->off_off, (size_t)bc * nc_off * g.S_o);
    take(p->off_m, (size_t)bc * nc_m * g.S_o);
    if (!backward) {
      take(p->off_out, (size_t)bc * g.O * g.S_o);
    } else {
      take(p->off_go, (size_t)bc * g.O * g.S_o);
      take(p->off_gi, (size_t)bc * g.C * g.S_i);
      take(p->off_goff, (size_t)bc * nc_off * g.S_o);
      take(p->off_gm, (size_t)bc * nc_m * g.S_o);
      take(p->off_gw, (size_t)g.O * g.Cg * g.K);
      take(p->off_gb, (size_t)g.O);
    }
  }
  p->total = off;
  return true;
}

template <typename H>
__global__ __launch_bounds__(256) void widen_kernel(const H *__restrict__ src, float *__restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    dst[i] = ld(src + i);
}
template <typename H, bool ACCUM>
__global__ __launch_bounds__(256) void narrow_kernel(const float *__restrict__ src, H *__restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    st(dst + i, ACCUM ? ld(dst + i) + src[i] : src[i]);
}
int nblocks(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}
// 16-bit tensors the native kernels do not take (hp_supported) run through fp32 copies: fp16 and bf16
int widen(int dtype, const void *src, float *dst, int64_t n, hipStream_t s) {
  if (n == 0) return MDCONV_OK;
  if (dtype == MDCONV_BF16)
    hipLaunchKernelGGL(widen_kernel<bf16_t>, dim3(nblocks(n)), dim3(256), 0, s, (const bf16_t *)src, dst, n);
  else
    hipLaunchKernelGGL(widen_kernel<__half>, dim3(nblocks(n)), dim3(256), 0, s, (const __half *)src, dst, n);
  return check_launch("widen");
}
template <typename H> void narrow_t(const float *src, void *dst, int64_t n, bool accum, hipStream_t s) {
  if (accum)
    hipLaunchKernelGGL((narrow_kernel<H, true>), dim3(nblocks(n)), dim3(256), 0, s, src, (H *)dst, n);
  else
    hipLaunchKernelGGL((narrow_kernel<H, false>), dim3(nblocks(n)), dim3(256), 0, s, src, (H *)dst, n);
}
int narrow(int dtype, const float *src, void *dst, int64_t n, bool accum, hipStream_t s) {
  if (n == 0) return MDCONV_OK;
  if (dtype == MDCONV_BF16) narrow_t<bf16_t>(src, dst, n, accum, s);
  else narrow_t<__half>(src, dst, n, accum, s);
  return check_launch("narrow");
}

// Fork / join helper for the one piece of the backward that does not depend on its neighbour: the
// grad_input gather (CSR build + col2im, HBM-bound) needs GEMM-1's grad_col and counters only, GEMM-2
// (matrix-bound) needs GEMM-1's packed grad_out and tap table only.  One side stream and two events
// per (device, caller stream), created on first use and kept (bounded like the weights-ready events).
struct Fork { hipStream_t side; hipEvent_t fork, join, bias; };
std::mutex g_fork_mu;
std::vector<std::pair<std::pair<int, hipStream_t>, Fork>> g_forks;
bool get_fork(hipStream_t stream, Fork *out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  std::lock_guard<std::mutex> lock(g_fork_mu);
  for (auto &e : g_forks)
    if (e.first.first == dev && e.first.second == stream) { *out = e.second; return true; }
  // Entries are never destroyed: another host thread may be between fork and join on any of them (advisor,
  // round 3).  Past 64 distinct (device, stream) callers a new one simply runs its backward unforked.
  if (g_forks.size() >= 64) return false;
  Fork f;
  // A stream of ANOTHER priority class: HIP multiplexes the streams of one class over a few hardware queues,
  // and once a process holds more streams (RCCL's, after init_process_group) the side stream can land on the
  // caller's queue -- the two tails then run one after the other again (measured: 3.44 -> 3.65 ms per cfg2
  // step under torchrun).  Priority classes have their own queues.
  int least = 0, greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
  if (hipStreamCreateWithPriority(&f.side, hipStreamNonBlocking, greatest) != hipSuccess) return false;
  if (hipEventCreateWithFlags(&f.fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&f.join, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&f.bias, hipEventDisableTiming) != hipSuccess)
    return false;
  g_forks.push_back({{dev, stream}, f});
  *out = f;
  return true;
}
// MDCONV_BWD_FORK = 0 / 1: run the grad_input gather beside GEMM-2 on a forked stream (read once).
// On by default: cfg2 3.68 -> 3.40 ms per step (GEMM-2 1.06 -> 1.10 ms, the 0.33 ms of CSR build +
// gather disappear under it); round 1 had measured a loss with the NCHW GEMM-2, whose gathers kept
// the L2 as busy as the col2im gather does -- the channels-last GEMM-2 leaves the L2 mostly idle.
int bwd_fork_mode() {
  static const int on = getenv("MDCONV_BWD_FORK") ? atoi(getenv("MDCONV_BWD_FORK")) : 1;
  return on;
}
bool bwd_fork_enabled() { return bwd_fork_mode() != 0; }

// fp32 backward of one chunk (all kernels accumulate into the grad_* pointers of `t`).
// Order on the caller's stream:
//   pack_wq, zero counters
//   -> GEMM-1 (+ coordinate gradients, grad_col, and for GEMM-2: packed grad_out, tap table,
//      grad_bias partials; CSR counting pass)
//   -> GEMM-2, split-K reduce, grad_bias -> [weights-ready event] -> CSR scan + fill -> col2im
// grad_weight / grad_bias are produced BEFORE the grad_input gather so that a data-parallel
// all-reduce of them can run under the gather (mdconv_stream_wait_weight_ready).
// The grad_input gather (CSR scan + fill -> col2im, HBM-bound) shares nothing with GEMM-2 (matrix-bound)
// and runs beside it on a forked stream that re-joins before this function returns (get_fork).
int backward_chunk_f32(const Geom &g, const Tensors &t, char *base, hipStream_t stream,
                       bool weights_final) {
  const BwdDims bd = bwd_dims(g);
  float *wq = (float *)(base + bd.off_wq);
  float *ga = (float *)(base + bd.off_ga);
  int *table = (int *)(base + bd.off_table);
  float *part = (float *)(base + bd.off_part);
  float *gcol = (float *)(base + bd.off_gcol);
  int *cnt = (int *)(base + bd.off_cnt), *rowptr = (int *)(base + bd.off_rowptr);
  void *entries = base + bd.off_entries;
  int rc;
  float *bias_part = g.with_bias ? (float *)(base + bd.off_bias) : nullptr;
  float *bstage = g.with_bias ? (float *)(base + bd.off_bstage) : nullptr;
  // channels-last copy of the input for the 3-D gathers of GEMM-1's drain and of GEMM-2
  float *xt = bd.cl ? (float *)(base + bd.off_xt) : nullptr;
  // pack_wq, counter clearing and the layout pass: one launch
  if ((rc = bwd_prep_f32(g, bd, (const float *)t.weight, wq, cnt, (const float *)t.input, xt, stream))) return rc;
  profile_mark(1, true, stream, "mfma_bwd_data_kernel");
  rc = mfma_bwd_data_f32(g, bd, t, wq, gcol, ga, bias_part, cnt, table, xt, stream);
  profile_mark(1, false, stream);
  if (rc) return rc;
  Fork fk;
  const bool fork = bwd_fork_enabled() && get_fork(stream, &fk);
  hipStream_t gs = stream;   // stream of the grad_input gather
  // GEMM-2 + split-K reduction; grad_bias behind them unless the forked stream already took it
  auto gemm2 = [&](bool bias_here) {
    int r = mfma_bwd_weight_f32(g, bd, t, ga, table, part, bias_part, xt, stream);
    if (!r && bias_here) r = grad_bias_f32(g, bd, bias_part, bstage, (float *)t.grad_bias, stream);
    if (!r && !bias_here && g.with_bias && hipStreamWaitEvent(stream, fk.bias, 0) != hipSuccess) {
      set_error("backward fork failed");
      r = MDCONV_ELAUNCH;
    }
    if (!r && weights_final) r = record_weight_ready(stream);
    return r;
  };
  if (fork) {
    if (hipEventRecord(fk.fork, stream) != hipSuccess || hipStreamWaitEvent(fk.side, fk.fork, 0) != hipSuccess) {
      set_error("backward fork failed");
      return MDCONV_ELAUNCH;
    }
    gs = fk.side;
  } else {
    if ((rc = gemm2(true))) return rc;
  }
  const bool gemm2_first = fork && bwd_fork_mode() == 2;   // (experiment: GEMM-2 enqueued before the gather)
  rc = MDCONV_OK;
  // forked: grad_bias first on the side stream (beside GEMM-2, off the critical path), its event for the caller's stream
  if (fork && g.with_bias) {
    rc = grad_bias_f32(g, bd, bias_part, bstage, (float *)t.grad_bias, gs);
    if (!rc && hipEventRecord(fk.bias, gs) != hipSuccess) { set_error("backward fork failed"); rc = MDCONV_ELAUNCH; }
  }
  if (!rc && gemm2_first) rc = gemm2(false);
  if (!rc) rc = csr_build_f32(g, bd, t, cnt, rowptr, entries, gs);
  if (!rc) {
    profile_mark(3, true, gs, bd.sample_keyed ? (bd.two_pass ? "col2im3d_sums_kernel" : "col2im3d_kernel") : "col2im_gather_kernel");
    rc = col2im_f32(g, bd, t, gcol, rowptr, entries, (float *)(base + bd.off_sums), gs);
    profile_mark(3, false, gs);
  }
  if (fork) {
    if (!rc && !gemm2_first) rc = gemm2(false);
    // join on every path after the fork (error returns included): the side stream must not outlive the call
    if ((hipEventRecord(fk.join, fk.side) != hipSuccess || hipStreamWaitEvent(stream, fk.join, 0) != hipSuccess) && !rc) {
      set_error("backward join failed");
      rc = MDCONV_ELAUNCH;
    }
  }
  return rc;
}

}  // namespace

// fork / join for the other kernel families (hp_host.hip): side stream that waits for everything enqueued on
// `stream` so far, or nullptr when forking is off or unavailable; join makes `stream` wait for the side stream
hipStream_t fork_side_stream(hipStream_t stream) {
  Fork fk;
  if (!bwd_fork_enabled() || !get_fork(stream, &fk)) return nullptr;
  if (hipEventRecord(fk.fork, stream) != hipSuccess || hipStreamWaitEvent(fk.side, fk.fork, 0) != hipSuccess) return nullptr;
  return fk.side;
}
int join_side_stream(hipStream_t stream) {
  Fork fk;
  if (!get_fork(stream, &fk)) return MDCONV_ELAUNCH;
  if (hipEventRecord(fk.join, fk.side) != hipSuccess || hipStreamWaitEvent(stream, fk.join, 0) != hipSuccess) {
    set_error("backward join failed");
    return MDCONV_ELAUNCH;
  }
  return MDCONV_OK;
}

// ---------------------------------------------------------------------------------------------
// 16-bit tensors on the shape-generic backward: it scatters grad_input / grad_weight with atomics,
// and a 16-bit atomic rounds at EVERY add (bf16: 2^-9 each).  So the call runs on fp32 copies in
// the workspace -- fresh, zeroed gradient buffers -- and each gradient is rounded once on the way out.
// ---------------------------------------------------------------------------------------------
namespace {
struct D16Plan { size_t off_x, off_off, off_m, off_w, off_go, off_gi, off_goff, off_gm, off_gw, off_gb, total; };
D16Plan direct16_plan(const Geom &g) {
  D16Plan p;
  size_t off = 0;
  auto take = [&](size_t &slot, size_t elems) { slot = off; off += align_up(elems * sizeof(float)); };
  const size_t n_x = (size_t)g.B * g.C * g.S_i, n_off = (size_t)g.B * g.DG * g.nd * g.K * g.S_o;
  const size_t n_m = (size_t)g.B * g.DG * g.K * g.S_o, n_w = (size_t)g.O * g.Cg * g.K, n_go = (size_t)g.B * g.O * g.S_o;
  take(p.off_x, n_x); take(p.off_off, n_off); take(p.off_m, n_m); take(p.off_w, n_w); take(p.off_go, n_go);
  take(p.off_gi, n_x); take(p.off_goff, n_off); take(p.off_gm, n_m); take(p.off_gw, n_w); take(p.off_gb, g.O);
  p.total = off;
  return p;
}
}  // namespace

size_t direct16_workspace_bytes(const Geom &g) { return direct16_plan(g).total; }

int direct16_backward(const Geom &g, int dtype, const Tensors &t, void *ws, hipStream_t stream) {
  const D16Plan p = direct16_plan(g);
  char *base = (char *)ws;
  const int64_t n_x = (int64_t)g.B * g.C * g.S_i, n_off = (int64_t)g.B * g.DG * g.nd * g.K * g.S_o;
  const int64_t n_m = (int64_t)g.B * g.DG * g.K * g.S_o, n_w = (int64_t)g.O * g.Cg * g.K, n_go = (int64_t)g.B * g.O * g.S_o;
  int rc;
  if ((rc = widen(dtype, t.input, (float *)(base + p.off_x), n_x, stream))) return rc;
  if ((rc = widen(dtype, t.offset, (float *)(base + p.off_off), n_off, stream))) return rc;
  if (t.mask && (rc = widen(dtype, t.mask, (float *)(base + p.off_m), n_m, stream))) return rc;
  if ((rc = widen(dtype, t.weight, (float *)(base + p.off_w), n_w, stream))) return rc;
  if ((rc = widen(dtype, t.grad_output, (float *)(base + p.off_go), n_go, stream))) return rc;
  if ((rc = zero_bytes(base + p.off_gi, p.total - p.off_gi, stream))) return rc;   // the five gradient buffers are contiguous
  Tensors tc = t;
  tc.input = base + p.off_x; tc.offset = base + p.off_off; tc.mask = t.mask ? base + p.off_m : nullptr;
  tc.weight = base + p.off_w; tc.grad_output = base + p.off_go;
  tc.grad_input = base + p.off_gi; tc.grad_offset = base + p.off_goff;
  tc.grad_mask = t.grad_mask ? base + p.off_gm : nullptr;
  tc.grad_weight = base + p.off_gw; tc.grad_bias = base + p.off_gb;
  Geom gc = g;
  gc.acc_data = gc.acc_w = 1;   // the kernels add into the zeroed fp32 buffers
  if ((rc = direct_backward(gc, MDCONV_F32, tc, stream))) return rc;
  if ((rc = narrow(dtype, (const float *)tc.grad_input, t.grad_input, n_x, g.acc_data != 0, stream))) return rc;
  if ((rc = narrow(dtype, (const float *)tc.grad_offset, t.grad_offset, n_off, g.acc_data != 0, stream))) return rc;
  if (t.grad_mask && (rc = narrow(dtype, (const float *)tc.grad_mask, t.grad_mask, n_m, g.acc_data != 0, stream))) return rc;
  if ((rc = narrow(dtype, (const float *)tc.grad_weight, t.grad_weight, n_w, g.acc_w != 0, stream))) return rc;
  if (g.with_bias && (rc = narrow(dtype, (const float *)tc.grad_bias, t.grad_bias, g.O, g.acc_w != 0, stream))) return rc;
  return MDCONV_OK;
}

static bool native_supported(const Geom &g, int dtype, bool backward) {
  if (dtype != MDCONV_F32 && dtype != MDCONV_F16 && dtype != MDCONV_BF16) return false;
  if (g.in_sz[g.nd - 1] < 2) return false;   // paired-corner gathers need 2 columns
  if (!backward) {
    if (g.Cg < 16 || g.Og < 16) return false;  // MFMA tiles would be mostly padding
    if (!(g.DG == 1 || (g.Cdg % (2 * kBK) == 0 && g.Cg % (2 * kBK) == 0))) return false;
  } else {
    // conv groups run as a block-diagonal dense weight (GEMM-1 skips the empty o-chunks, GEMM-2
    // the empty waves); deformable groups must be whole 64-channel blocks
    if (g.C < 16 || g.O < 16 || g.C % 8) return false;
    if (!(g.DG == 1 || g.Cdg == 64 || g.Cdg == 128 || g.Cdg % 256 == 0)) return false;
  }
  if (backward && bwd_data_lds_bytes(g, bwd_dims(g)) > kBwdDataLdsCap) return false;   // grad_out tile lives in LDS
  Plan p;
  return make_plan(g, dtype, backward, &p);   // one image must fit 32-bit buffer offsets
}

static size_t native_workspace_bytes(const Geom &g, int dtype, bool backward) {
  Plan p;
  if (!make_plan(g, dtype, backward, &p)) return 0;
  return p.total;
}

static int native_forward(const Geom &g, int dtype, const Tensors &t, void *ws, hipStream_t stream) {
  Plan p;
  if (!make_plan(g, dtype, false, &p)) { set_error("mfma_forward: no plan"); return MDCONV_EUNSUPPORTED; }
  char *base = (char *)ws;
  const size_t es = p.half_io ? 2 : 4;
  const int nc_off = g.DG * g.nd * g.K, nc_m = g.DG * g.K;
  int rc;
  const float *w32 = (const float *)t.weight, *b32 = (const float *)t.bias;
  if (p.half_io) {
    if ((rc = widen(dtype, t.weight, (float *)(base + p.off_w), (int64_t)g.O * g.Cg * g.K, stream))) return rc;
    if (g.with_bias && (rc = widen(dtype, t.bias, (float *)(base + p.off_b), g.O, stream))) return rc;
    w32 = (const float *)(base + p.off_w);
    b32 = (const float *)(base + p.off_b);
  }
  const PackDims pd = pack_dims(p.gc);
  float *wp = (float *)base;
  if ((rc = pack_weights_f32(p.gc, pd, w32, wp, nullptr, stream))) return rc;
  for (int b0 = 0; b0 < g.B; b0 += p.Bc) {
    const int bc = g.B - b0 < p.Bc ? g.B - b0 : p.Bc;
    const Geom gc = chunk_geom(g, bc);
    Tensors tc = {};
    const char *x = (const char *)t.input + (size_t)b0 * g.C * g.S_i * es;
    const char *of = (const char *)t.offset + (size_t)b0 * nc_off * g.S_o * es;
    const char *mk = t.mask ? (const char *)t.mask + (size_t)b0 * nc_m * g.S_o * es : nullptr;
    char *out = (char *)t.output + (size_t)b0 * g.O * g.S_o * es;
    tc.weight = w32;
    tc.bias = b32;
    if (p.half_io) {
      if ((rc = widen(dtype, x, (float *)(base + p.off_x), (int64_t)bc * g.C * g.S_i, stream))) return rc;
      if ((rc = widen(dtype, of, (float *)(base + p.off_off), (int64_t)bc * nc_off * g.S_o, stream))) return rc;
      if (mk && (rc = widen(dtype, mk, (float *)(base + p.off_m), (int64_t)bc * nc_m * g.S_o, stream))) return rc;
      tc.input = base + p.off_x;
      tc.offset = base + p.off_off;
      tc.mask = mk ? base + p.off_m : nullptr;
      tc.output = base + p.off_out;
    } else {
      tc.input = x; tc.offset = of; tc.mask = mk; tc.output = out;
    }
    profile_mark(0, true, stream, fwd_channels_last(gc) ? "mfma_fwd_cl_kernel" : "mfma_fwd_kernel");
    if (fwd_channels_last(gc)) {
      float *xt = (float *)(base + align_up((size_t)gc.G * gc.K * pd.Cgp * pd.Ogp * sizeof(float)));
      float *part = (float *)((char *)xt + align_up(fwd_cl_bytes(gc)));
      rc = mfma_forward_cl_f32(gc, pd, tc, wp, fwd_tail_bytes(gc) ? part : nullptr, xt, stream);
    } else {
      float *part = (float *)(base + align_up((size_t)gc.G * gc.K * pd.Cgp * pd.Ogp * sizeof(float)));
      rc = mfma_forward_f32(gc, pd, tc, wp, fwd_tail_bytes(gc) ? part : nullptr, stream);
    }
    profile_mark(0, false, stream);
    if (rc) return rc;
    if (p.half_io && (rc = narrow(dtype, (const float *)tc.output, out, (int64_t)bc * g.O * g.S_o, false, stream)))
      return rc;
  }
  return MDCONV_OK;
}

static int native_backward(const Geom &g, int dtype, const Tensors &t, void *ws, hipStream_t stream) {
  Plan p;
  if (!make_plan(g, dtype, true, &p)) { set_error("mfma_backward: no plan"); return MDCONV_EUNSUPPORTED; }
  char *base = (char *)ws;
  const size_t es = p.half_io ? 2 : 4;
  const int nc_off = g.DG * g.nd * g.K, nc_m = g.DG * g.K;
  const int64_t n_w = (int64_t)g.O * g.Cg * g.K;
  int rc;
  if (p.half_io) {
    if ((rc = widen(dtype, t.weight, (float *)(base + p.off_w), n_w, stream))) return rc;
  }
  for (int b0 = 0; b0 < g.B; b0 += p.Bc) {
    const int bc = g.B - b0 < p.Bc ? g.B - b0 : p.Bc;
    Geom gc = chunk_geom(g, bc);
    // grad_weight / grad_bias: chunks after the first always add; the fp32 temporaries of the fp16
    // path are fresh memory, so every kernel overwrites them (no zero fills) and the caller's mode
    // is applied when they are narrowed back
    gc.acc_w = (b0 > 0) ? 1 : (p.half_io ? 0 : g.acc_w);
    gc.acc_data = p.half_io ? 0 : g.acc_data;
    const size_t o_x = (size_t)b0 * g.C * g.S_i, o_off = (size_t)b0 * nc_off * g.S_o;
    const size_t o_m = (size_t)b0 * nc_m * g.S_o, o_go = (size_t)b0 * g.O * g.S_o;
    Tensors tc = t;
    if (p.half_io) {
      const int64_t n_x = (int64_t)bc * g.C * g.S_i, n_off = (int64_t)bc * nc_off * g.S_o;
      const int64_t n_m = (int64_t)bc * nc_m * g.S_o, n_go = (int64_t)bc * g.O * g.S_o;
      if ((rc = widen(dtype, (const char *)t.input + o_x * es, (float *)(base + p.off_x), n_x, stream))) return rc;
      if ((rc = widen(dtype, (const char *)t.offset + o_off * es, (float *)(base + p.off_off), n_off, stream))) return rc;
      if (t.mask && (rc = widen(dtype, (const char *)t.mask + o_m * es, (float *)(base + p.off_m), n_m, stream))) return rc;
      if ((rc = widen(dtype, (const char *)t.grad_output + o_go * es, (float *)(base + p.off_go), n_go, stream))) return rc;
      tc.input = base + p.off_x; tc.offset = base + p.off_off; tc.mask = t.mask ? base + p.off_m : nullptr;
      tc.weight = base + p.off_w; tc.grad_output = base + p.off_go;
      tc.grad_input = base + p.off_gi; tc.grad_offset = base + p.off_goff;
      tc.grad_mask = t.grad_mask ? base + p.off_gm : nullptr;
      tc.grad_weight = base + p.off_gw; tc.grad_bias = base + p.off_gb;
      if ((rc = backward_chunk_f32(gc, tc, base, stream, false))) return rc;
      if ((rc = narrow(dtype, (const float *)tc.grad_input, (char *)t.grad_input + o_x * es, n_x, g.acc_data != 0, stream))) return rc;
      if ((rc = narrow(dtype, (const float *)tc.grad_offset, (char *)t.grad_offset + o_off * es, n_off, g.acc_data != 0, stream))) return rc;
      if (t.grad_mask &&
          (rc = narrow(dtype, (const float *)tc.grad_mask, (char *)t.grad_mask + o_m * es, n_m, g.acc_data != 0, stream)))
        return rc;
    } else {
      tc.input = (const char *)t.input + o_x * es;
      tc.offset = (const char *)t.offset + o_off * es;
      tc.mask = t.mask ? (const char *)t.mask + o_m * es : nullptr;
      tc.grad_output = (const char *)t.grad_output + o_go * es;
      tc.grad_input = (char *)t.grad_input + o_x * es;
      tc.grad_offset = (char *)t.grad_offset + o_off * es;
      tc.grad_mask = t.grad_mask ? (char *)t.grad_mask + o_m * es : nullptr;
      if ((rc = backward_chunk_f32(gc, tc, base, stream, b0 + bc >= g.B))) return rc;
    }
  }
  if (p.half_io) {
    if ((rc = narrow(dtype, (const float *)(base + p.off_gw), t.grad_weight, n_w, g.acc_w != 0, stream))) return rc;
    if (g.with_bias && (rc = narrow(dtype, (const float *)(base + p.off_gb), t.grad_bias, g.O, g.acc_w != 0, stream))) return rc;
    if ((rc = record_weight_ready(stream))) return rc;
  }
  return MDCONV_OK;
}

// ---------------------------------------------------------------------------------------------
// Backward for deformable groups the kernels above do not tile (C_in / DG of 16, 24, 32, 40 ...): the
// gradients of deformable group dg involve its own input channels, offsets and masks and nothing of the
// other groups (mdeformable_conv.cu:231 indexes the offsets by c / channel_per_deformable_group), so the
// call is DG independent single-group problems over channel slices -- each copied into the workspace
// (strided 2-D copies, a few % of the kernels' traffic), run through the same matrix-core pipeline and
// copied back.  Slower per sample than a native tiling (C_in / DG = 32 fills half of a 64-channel tile)
// but an order of magnitude faster than the shape-generic scatter kernels these shapes used to reach.
// ---------------------------------------------------------------------------------------------
namespace {
struct SplitPlan {
  Geom gs;            // one slice: DG = 1, C = C_in / DG, the conv groups / output channels it touches
  bool copy_w, copy_go;
  size_t off_x, off_off, off_m, off_go, off_w, off_gi, off_goff, off_gm, off_gw, off_sub, sub_bytes, total;
};
bool split_slice_geom(const Geom &g, Geom *out, bool *copy_w, bool *copy_go) {
  if (g.DG <= 1 || g.Cdg < 16 || g.Cdg % 8) return false;
  Geom s = g;
  s.DG = 1; s.C = g.Cdg; s.Cdg = g.Cdg; s.with_bias = 0;
  if (g.Cg % g.Cdg == 0) {          // the slice lies inside one conv group
    s.G = 1; s.Cg = g.Cdg; s.O = s.Og = g.Og;
    *copy_w = g.Cg != g.Cdg;
  } else if (g.Cdg % g.Cg == 0) {   // the slice is a run of whole conv groups
    s.G = g.Cdg / g.Cg; s.Cg = g.Cg; s.Og = g.Og; s.O = s.G * g.Og;
    *copy_w = false;
  } else {
    return false;
  }
  *copy_go = s.O != g.O;
  *out = s;
  return true;
}
bool split_plan(const Geom &g, int dtype, SplitPlan *p) {
  if (!split_slice_geom(g, &p->gs, &p->copy_w, &p->copy_go)) return false;
  if (!native_supported(p->gs, dtype, true)) return false;
  const size_t es = dtype == MDCONV_F32 ? 4 : 2;
  const Geom &s = p->gs;
  size_t off = 0;
  auto take = [&](size_t &slot, size_t elems) { slot = off; off += align_up(elems * es); };
  take(p->off_x, (size_t)g.B * s.C * g.S_i);
  take(p->off_off, (size_t)g.B * g.nd * g.K * g.S_o);
  take(p->off_m, g.modulated ? (size_t)g.B * g.K * g.S_o : 0);
  take(p->off_go, p->copy_go ? (size_t)g.B * s.O * g.S_o : 0);
  take(p->off_w, p->copy_w ? (size_t)s.O * s.Cg * g.K : 0);
  take(p->off_gi, (size_t)g.B * s.C * g.S_i);
  take(p->off_goff, (size_t)g.B * g.nd * g.K * g.S_o);
  take(p->off_gm, g.modulated ? (size_t)g.B * g.K * g.S_o : 0);
  take(p->off_gw, p->copy_w ? (size_t)s.O * s.Cg * g.K : 0);
  p->off_sub = off;
  // sized for the slice WITH bias: the first slice of a conv group runs with with_bias = 1 (split_backward) and then
  // has the grad_bias stage buffer at the end of its layout -- sized without it, that slice wrote 32 * C_out * 4 bytes
  // past the workspace (found by tools/fuzz_more.py in round 5; tests/test_gpu_workspace_guard.py)
  Geom sb = s;
  sb.with_bias = g.with_bias;
  p->sub_bytes = native_workspace_bytes(sb, dtype, true);
  p->total = off + p->sub_bytes;
  return true;
}
// strided row copy as a KERNEL, not hipMemcpy2DAsync: memcpy / memset nodes made HIP graph replay fault
// (see zero_bytes), and the library promises plain kernel sequences that capture cleanly.  Pitches and widths are
// multiples of 2 bytes (element sizes 2 / 4), 4-byte words where everything is 4-byte aligned.
template <typename W>
__global__ __launch_bounds__(256) void copy_rows_kernel(W *__restrict__ dst, int64_t dpitch, const W *__restrict__ src,
                                                        int64_t spitch, int64_t width, int64_t rows) {
  const int64_t n = width * rows;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / width, c = i - r * width;
    dst[r * dpitch + c] = src[r * spitch + c];
  }
}
int copy_rows(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t rows, hipStream_t stream) {
  if (width == 0 || rows == 0) return MDCONV_OK;
  const bool words = ((dpitch | spitch | width) & 3) == 0 && (((uintptr_t)dst | (uintptr_t)src) & 3) == 0;
  const size_t es = words ? 4 : 2;
  const int64_t n = (int64_t)(width / es) * (int64_t)rows;
  const int64_t blocks = (n + 255) / 256;
  const dim3 grid((unsigned)(blocks > 16384 ? 16384 : blocks));
  if (words)
    hipLaunchKernelGGL(copy_rows_kernel<unsigned>, grid, dim3(256), 0, stream, (unsigned *)dst, (int64_t)(dpitch / 4),
                       (const unsigned *)src, (int64_t)(spitch / 4), (int64_t)(width / 4), (int64_t)rows);
  else
    hipLaunchKernelGGL(copy_rows_kernel<unsigned short>, grid, dim3(256), 0, stream, (unsigned short *)dst,
                       (int64_t)(dpitch / 2), (const unsigned short *)src, (int64_t)(spitch / 2), (int64_t)(width / 2),
                       (int64_t)rows);
  return check_launch("copy_rows");
}

int split_backward(const Geom &g, int dtype, const SplitPlan &p, const Tensors &t, void *ws, hipStream_t stream) {
  char *base = (char *)ws;
  const size_t es = dtype == MDCONV_F32 ? 4 : 2;
  const Geom &s = p.gs;
  const size_t w_x = (size_t)s.C * g.S_i * es, p_x = (size_t)g.C * g.S_i * es;
  const size_t w_off = (size_t)g.nd * g.K * g.S_o * es, p_off = w_off * g.DG;
  const size_t w_m = (size_t)g.K * g.S_o * es, p_m = w_m * g.DG;
  const size_t w_go = (size_t)s.O * g.S_o * es, p_go = (size_t)g.O * g.S_o * es;
  const size_t w_w = (size_t)s.Cg * g.K * es, p_w = (size_t)g.Cg * g.K * es;
  int rc;
  for (int dg = 0; dg < g.DG; ++dg) {
    const int c0 = dg * g.Cdg;            // first input channel of the slice
    const int grp = c0 / g.Cg;            // first conv group it touches
    const int o0 = grp * g.Og;            // first output channel of those groups
    const int cw = c0 - grp * g.Cg;       // channel offset inside the group's weight rows
    Tensors ts = t;
    const char *src_x = (const char *)t.input + (size_t)c0 * g.S_i * es;
    const char *src_off = (const char *)t.offset + (size_t)dg * w_off;
    char *dst_gi = (char *)t.grad_input + (size_t)c0 * g.S_i * es;
    char *dst_goff = (char *)t.grad_offset + (size_t)dg * w_off;
    char *dst_gw = (char *)t.grad_weight + ((size_t)o0 * g.Cg + cw) * g.K * es;
    if ((rc = copy_rows(base + p.off_x, w_x, src_x, p_x, w_x, g.B, stream))) return rc;
    if ((rc = copy_rows(base + p.off_off, w_off, src_off, p_off, w_off, g.B, stream))) return rc;
    ts.input = base + p.off_x; ts.offset = base + p.off_off;
    ts.grad_input = base + p.off_gi; ts.grad_offset = base + p.off_goff;
    if (g.modulated) {
      if ((rc = copy_rows(base + p.off_m, w_m, (const char *)t.mask + (size_t)dg * w_m, p_m, w_m, g.B, stream))) return rc;
      ts.mask = base + p.off_m; ts.grad_mask = base + p.off_gm;
    }
    ts.grad_output = (const char *)t.grad_output + (size_t)o0 * g.S_o * es;
    if (p.copy_go) {
      if ((rc = copy_rows(base + p.off_go, w_go, ts.grad_output, p_go, w_go, g.B, stream))) return rc;
      ts.grad_output = base + p.off_go;
    }
    ts.weight = (const char *)t.weight + ((size_t)o0 * g.Cg + cw) * g.K * es;
    ts.grad_weight = dst_gw;
    if (p.copy_w) {
      if ((rc = copy_rows(base + p.off_w, w_w, ts.weight, p_w, w_w, s.O, stream))) return rc;
      ts.weight = base + p.off_w; ts.grad_weight = base + p.off_gw;
    }
    Geom gs = s;
    // grad_bias belongs to the output channels: once per conv group, with the first slice that touches it
    gs.with_bias = g.with_bias && cw == 0 ? 1 : 0;
    ts.bias = nullptr;
    ts.grad_bias = gs.with_bias ? (char *)t.grad_bias + (size_t)o0 * es : nullptr;
    if (g.acc_data) {   // accumulate mode: the slice starts from the caller's values
      if ((rc = copy_rows(base + p.off_gi, w_x, dst_gi, p_x, w_x, g.B, stream))) return rc;
      if ((rc = copy_rows(base + p.off_goff, w_off, dst_goff, p_off, w_off, g.B, stream))) return rc;
      if (g.modulated &&
          (rc = copy_rows(base + p.off_gm, w_m, (const char *)t.grad_mask + (size_t)dg * w_m, p_m, w_m, g.B, stream)))
        return rc;
    }
    if (g.acc_w && p.copy_w && (rc = copy_rows(base + p.off_gw, w_w, dst_gw, p_w, w_w, s.O, stream))) return rc;
    if ((rc = native_backward(gs, dtype, ts, base + p.off_sub, stream))) return rc;
    if ((rc = copy_rows(dst_gi, p_x, base + p.off_gi, w_x, w_x, g.B, stream))) return rc;
    if ((rc = copy_rows(dst_goff, p_off, base + p.off_goff, w_off, w_off, g.B, stream))) return rc;
    if (g.modulated &&
        (rc = copy_rows((char *)t.grad_mask + (size_t)dg * w_m, p_m, base + p.off_gm, w_m, w_m, g.B, stream)))
      return rc;
    if (p.copy_w && (rc = copy_rows(dst_gw, p_w, base + p.off_gw, w_w, w_w, s.O, stream))) return rc;
  }
  return record_weight_ready(stream);   // after the last slice's copies
}
}  // namespace

// Forward of the same shapes: the slices of one conv group add up in its output channels, so each slice's
// output goes to a workspace tile and is copied (first slice of the conv group: it carries the bias) or added
// (fp32 only: adding rounded 16-bit outputs would round DG times) into the caller's rows.
namespace {
__global__ __launch_bounds__(256) void add_rows_kernel(float *__restrict__ dst, int64_t dpitch,
                                                       const float *__restrict__ src, int64_t width, int64_t rows) {
  const int64_t n = width * rows;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / width;
    dst[r * dpitch + (i - r * width)] += src[i];
  }
}
struct SplitFwdPlan {
  Geom gs;
  bool copy_w, copy_out;
  size_t off_x, off_off, off_m, off_w, off_out, off_sub, total;
};
bool split_fwd_plan(const Geom &g, int dtype, SplitFwdPlan *p) {
  if (!split_slice_geom(g, &p->gs, &p->copy_w, &p->copy_out)) return false;
  if (g.Cg > g.Cdg && dtype != MDCONV_F32) return false;   // slices of one conv group are summed: fp32 only
  // narrow conv groups are cheap on the shape-generic forward (C=128, 8 groups of 16, DG=4, 64x64, B=16: 0.3 ms
  // faster there than as four slices with mostly-padding tiles); wide ones are not (one group, DG=8: 0.25 ms slower)
  if (g.Cg < 64) return false;
  if (!native_supported(p->gs, dtype, false)) return false;
  const size_t es = dtype == MDCONV_F32 ? 4 : 2;
  const Geom &s = p->gs;
  size_t off = 0;
  auto take = [&](size_t &slot, size_t elems) { slot = off; off += align_up(elems * es); };
  take(p->off_x, (size_t)g.B * s.C * g.S_i);
  take(p->off_off, (size_t)g.B * g.nd * g.K * g.S_o);
  take(p->off_m, g.modulated ? (size_t)g.B * g.K * g.S_o : 0);
  take(p->off_w, p->copy_w ? (size_t)s.O * s.Cg * g.K : 0);
  take(p->off_out, (size_t)g.B * s.O * g.S_o);
  p->off_sub = off;
  p->total = off + native_workspace_bytes(s, dtype, false);
  return true;
}
int split_forward(const Geom &g, int dtype, const SplitFwdPlan &p, const Tensors &t, void *ws, hipStream_t stream) {
  char *base = (char *)ws;
  const size_t es = dtype == MDCONV_F32 ? 4 : 2;
  const Geom &s = p.gs;
  const size_t w_x = (size_t)s.C * g.S_i * es, p_x = (size_t)g.C * g.S_i * es;
  const size_t w_off = (size_t)g.nd * g.K * g.S_o * es, p_off = w_off * g.DG;
  const size_t w_m = (size_t)g.K * g.S_o * es, p_m = w_m * g.DG;
  const size_t w_out = (size_t)s.O * g.S_o * es, p_out = (size_t)g.O * g.S_o * es;
  const size_t w_w = (size_t)s.Cg * g.K * es, p_w = (size_t)g.Cg * g.K * es;
  int rc;
  for (int dg = 0; dg < g.DG; ++dg) {
    const int c0 = dg * g.Cdg, grp = c0 / g.Cg, o0 = grp * g.Og, cw = c0 - grp * g.Cg;
    Tensors ts = t;
    if ((rc = copy_rows(base + p.off_x, w_x, (const char *)t.input + (size_t)c0 * g.S_i * es, p_x, w_x, g.B, stream))) return rc;
    if ((rc = copy_rows(base + p.off_off, w_off, (const char *)t.offset + (size_t)dg * w_off, p_off, w_off, g.B, stream))) return rc;
    ts.input = base + p.off_x; ts.offset = base + p.off_off;
    if (g.modulated) {
      if ((rc = copy_rows(base + p.off_m, w_m, (const char *)t.mask + (size_t)dg * w_m, p_m, w_m, g.B, stream))) return rc;
      ts.mask = base + p.off_m;
    }
    ts.weight = (const char *)t.weight + ((size_t)o0 * g.Cg + cw) * g.K * es;
    if (p.copy_w) {
      if ((rc = copy_rows(base + p.off_w, w_w, ts.weight, p_w, w_w, s.O, stream))) return rc;
      ts.weight = base + p.off_w;
    }
    Geom gs = s;
    gs.with_bias = g.with_bias && cw == 0 ? 1 : 0;
    ts.bias = gs.with_bias ? (const char *)t.bias + (size_t)o0 * es : nullptr;
    ts.output = base + p.off_out;
    if ((rc = native_forward(gs, dtype, ts, base + p.off_sub, stream))) return rc;
    char *dst = (char *)t.output + (size_t)o0 * g.S_o * es;
    if (cw == 0) {
      if ((rc = copy_rows(dst, p_out, base + p.off_out, w_out, w_out, g.B, stream))) return rc;
    } else {
      const int64_t width = (int64_t)s.O * g.S_o, n = width * g.B;
      const int64_t blocks = (n + 255) / 256;
      hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, stream,
                         (float *)dst, (int64_t)g.O * g.S_o, (const float *)(base + p.off_out), width, (int64_t)g.B);
      if ((rc = check_launch("add_rows"))) return rc;
    }
  }
  return MDCONV_OK;
}
}  // namespace

// ---------------------------------------------------------------------------------------------
// The same shapes as ONE padded problem (round 6): every deformable group widened to a size the kernels tile (forward: whole
// 32-channel K stages; backward: 64 / 128 / n x 256 channels) with zero input planes and zero weight rows in between -- the
// padding channels add nothing to any output, and their own gradient rows are never copied back.  One launch sequence over
// C' = DG x padded-group channels instead of DG sequences over mostly-padding tiles plus their copies: faster on all 13 shapes
// measured, 4x growth included (fp32, 4 groups: 64 -> 64 at 56 x 56, B = 16 1.44 -> 1.00 ms; 192 -> 192 at 20 x 20 0.70 -> 0.30;
// 3-D 64 -> 64 2.41 -> 1.12; profiles/r06_experiments.md 18).  Taken when the padded problem is at most a few times
// the caller's (kPadMaxGrowth); one conv group only (conv groups keep the slices above).
// ---------------------------------------------------------------------------------------------
namespace {
// 16 -> 16 channels in 2 groups (8 -> 32 forward, 8 -> 64 backward): 0.39 ms on the shape-generic kernels, 0.25 padded; in 4 groups
// (4 -> 32 / 64) at 40 x 40, B = 8: 0.43 ms generic against 0.34 padded, and the generic kernels fall further behind with every output
// channel (16 -> 256 in 4 groups: 1.76 vs 0.64 ms; 3-D: 5.68 vs 1.27); groups of 2 channels (32x) lose at 16 output channels
// (profiles/r06_experiments.md 20, 24)
constexpr int kPadMaxGrowth = 16;
struct PadPlan {
  Geom gp;            // the padded problem
  bool pad_c, pad_o;  // input channels / output channels padded
  // channel groups of the input (conv groups, else deformable groups): count, channels each (caller's / padded);
  // output groups (conv groups): count, channels each; weight sub-rows per output channel ([O][DG][C_dg][K] with one conv group)
  int ng, cin, cinp, nog, og, ogp, wsub;
  size_t off_x, off_w, off_gi, off_gw, off_o, off_b, off_gb, off_sub, total;   // off_o: output (forward) / grad_output (backward)
};
// dst[r][0 .. dwidth) = src[r][0 .. width) followed by zeros (element = W)
template <typename W>
__global__ __launch_bounds__(256) void pad_rows_kernel(W *__restrict__ dst, int64_t dwidth, const W *__restrict__ src,
                                                       int64_t width, int64_t rows) {
  const int64_t n = dwidth * rows;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / dwidth, c = i - r * dwidth;
    dst[i] = c < width ? src[r * width + c] : (W)0;
  }
}
int pad_rows(void *dst, size_t dwidth, const void *src, size_t width, size_t rows, hipStream_t stream) {
  const bool words = ((dwidth | width) & 3) == 0 && (((uintptr_t)dst | (uintptr_t)src) & 3) == 0;
  const size_t es = words ? 4 : 2;
  const int64_t n = (int64_t)(dwidth / es) * (int64_t)rows;
  const int64_t blocks = (n + 255) / 256;
  const dim3 grid((unsigned)(blocks > 16384 ? 16384 : (blocks < 1 ? 1 : blocks)));
  if (words)
    hipLaunchKernelGGL(pad_rows_kernel<unsigned>, grid, dim3(256), 0, stream, (unsigned *)dst, (int64_t)(dwidth / 4),
                       (const unsigned *)src, (int64_t)(width / 4), (int64_t)rows);
  else
    hipLaunchKernelGGL(pad_rows_kernel<unsigned short>, grid, dim3(256), 0, stream, (unsigned short *)dst,
                       (int64_t)(dwidth / 2), (const unsigned short *)src, (int64_t)(width / 2), (int64_t)rows);
  return check_launch("pad_rows");
}
// Rows in groups of `inner` (padded: `inner_p`), `outer` groups: dst row (q, r) = src row (q, r) widened to dwidth with zeros for
// r < inner, a zero row for inner <= r < inner_p (weights: the rows of one conv group's output channels, padded to the kernels' floor)
template <typename W>
__global__ __launch_bounds__(256) void pad_rows_grouped_kernel(W *__restrict__ dst, int64_t dwidth, const W *__restrict__ src,
                                                               int64_t width, int64_t inner, int64_t inner_p, int64_t outer) {
  const int64_t n = dwidth * inner_p * outer;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t rd = i / dwidth, c = i - rd * dwidth;
    const int64_t q = rd / inner_p, r = rd - q * inner_p;
    dst[i] = (r < inner && c < width) ? src[(q * inner + r) * width + c] : (W)0;
  }
}
// the inverse: dst row (q, r) (width elements) = the first `width` elements of src row (q, r) of the padded layout
template <typename W>
__global__ __launch_bounds__(256) void unpad_rows_grouped_kernel(W *__restrict__ dst, int64_t width, const W *__restrict__ src,
                                                                 int64_t swidth, int64_t inner, int64_t inner_p, int64_t outer) {
  const int64_t n = width * inner * outer;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t rs = i / width, c = i - rs * width;
    const int64_t q = rs / inner, r = rs - q * inner;
    dst[i] = src[(q * inner_p + r) * swidth + c];
  }
}
int pad_rows_grouped(void *dst, size_t dwidth, const void *src, size_t width, size_t inner, size_t inner_p, size_t outer,
                     hipStream_t stream) {
  const bool words = ((dwidth | width) & 3) == 0 && (((uintptr_t)dst | (uintptr_t)src) & 3) == 0;
  const size_t es = words ? 4 : 2;
  const int64_t n = (int64_t)(dwidth / es) * (int64_t)(inner_p * outer);
  const int64_t blocks = (n + 255) / 256;
  const dim3 grid((unsigned)(blocks > 16384 ? 16384 : (blocks < 1 ? 1 : blocks)));
  if (words)
    hipLaunchKernelGGL(pad_rows_grouped_kernel<unsigned>, grid, dim3(256), 0, stream, (unsigned *)dst, (int64_t)(dwidth / 4),
                       (const unsigned *)src, (int64_t)(width / 4), (int64_t)inner, (int64_t)inner_p, (int64_t)outer);
  else
    hipLaunchKernelGGL(pad_rows_grouped_kernel<unsigned short>, grid, dim3(256), 0, stream, (unsigned short *)dst,
                       (int64_t)(dwidth / 2), (const unsigned short *)src, (int64_t)(width / 2), (int64_t)inner, (int64_t)inner_p,
                       (int64_t)outer);
  return check_launch("pad_rows_grouped");
}
int unpad_rows_grouped(void *dst, size_t width, const void *src, size_t swidth, size_t inner, size_t inner_p, size_t outer,
                       hipStream_t stream) {
  const bool words = ((swidth | width) & 3) == 0 && (((uintptr_t)dst | (uintptr_t)src) & 3) == 0;
  const size_t es = words ? 4 : 2;
  const int64_t n = (int64_t)(width / es) * (int64_t)(inner * outer);
  const int64_t blocks = (n + 255) / 256;
  const dim3 grid((unsigned)(blocks > 16384 ? 16384 : (blocks < 1 ? 1 : blocks)));
  if (words)
    hipLaunchKernelGGL(unpad_rows_grouped_kernel<unsigned>, grid, dim3(256), 0, stream, (unsigned *)dst, (int64_t)(width / 4),
                       (const unsigned *)src, (int64_t)(swidth / 4), (int64_t)inner, (int64_t)inner_p, (int64_t)outer);
  else
    hipLaunchKernelGGL(unpad_rows_grouped_kernel<unsigned short>, grid, dim3(256), 0, stream, (unsigned short *)dst,
                       (int64_t)(width / 2), (const unsigned short *)src, (int64_t)(swidth / 2), (int64_t)inner, (int64_t)inner_p,
                       (int64_t)outer);
  return check_launch("unpad_rows_grouped");
}
// MDCONV_DG_PLAN = pad | split forces one plan where both exist (developer A/B; default: by growth)
int dg_plan_env() {
  static const int v = [] {
    const char *e = getenv("MDCONV_DG_PLAN");
    return !e ? 0 : (!strcmp(e, "pad") ? 1 : (!strcmp(e, "split") ? 2 : 0));
  }();
  return v;
}
// ONE deformable group and one conv group: two kinds of shapes run as a padded problem although nothing about their groups
// needs it (profiles/r06_experiments.md 22, 23).
//  * C_in not a multiple of the 64-channel slab of the channels-last kernels.  Such shapes are tiled natively, but by the NCHW
//    kernels, whose 2^nd corner loads go to one channel PLANE each; padded to the next multiple of 64 they take the channels-last
//    kernels.  3-D from 2048 output pixels (32 -> 64 at 16 x 56 x 56, B = 2: 3.33 -> 1.80 ms; 16 -> 16 at 16 x 32 x 32: 0.98 -> 0.72;
//    160 channels at 1568 pixels: +7 %, hence the floor); 2-D only for 32 <= C_in < 64 from 8192 pixels, where the backward is
//    channels-last anyway (48 -> 48 at 56 x 56, B = 16: 0.41 -> 0.34 ms; 96 / 160 channels lose 10-15 %).
//  * Fewer than 16 input or output channels: below the matrix kernels' floor, i.e. the shape-generic kernels -- whose cost grows
//    with C_in x C_out x taps per thread.  Output channels are padded to 16 (zero weight rows, zero grad_output planes, a
//    workspace tile for the output), input channels to 64: 3-D 64 -> 8 at 8 x 28 x 28: 6.92 -> 0.45 ms, 2-D 64 -> 8 at 56 x 56,
//    B = 16: 3.08 -> 0.32 ms, 3-D 8 -> 8 at 16 x 32 x 32: 1.81 -> 0.73 ms, 2-D 8 -> 8 at 112 x 112, B = 8: 0.81 -> 0.57 ms.  Not for
//    grids of a few hundred pixels (4 -> 4 at 8 x 8, BASELINE configs[0]: 0.13 ms generic, 0.21 padded), nor in 2-D below 8 input
//    channels (3 -> 16 at 112 x 112: 0.38 -> 0.55 ms) or 8192 pixels (ties).
// MDCONV_PAD_CHANNELS = 0 | 1: never / wherever eligible (the test suite's way to reach the plan with small shapes).
bool pad_channels_preferred(const Geom &g) {
  static const int env = getenv("MDCONV_PAD_CHANNELS") ? atoi(getenv("MDCONV_PAD_CHANNELS")) : -1;
  if (env == 0 || g.DG != 1) return false;
  if (g.G != 1)   // conv groups: the 3-D slab rule per group (3-D 200 -> 64 in 2 groups at 8 x 20 x 20: 1.19 ms, 256 -> 64: 0.61)
    // (at most 2x: 64 -> 128 in 4 groups of 16 -> 64 at 8 x 14 x 14 lost 26 %)
    return g.nd == 3 && g.Cg >= 32 && g.Cg % 64 != 0 && (env > 0 || g.N >= 2048);
  const bool tiny_c = g.C < 16, tiny_o = g.O < 16;
  if (!tiny_c && !tiny_o && g.C % 64 == 0) return false;
  if (env > 0) return true;
  if (tiny_c) return g.nd == 3 ? g.N >= 512 : (g.C >= 8 && g.N >= 8192);
  if (tiny_o) return g.N >= 512;
  if (g.nd == 3) return g.N >= 2048;
  return g.C >= 32 && g.C < 64 && g.N >= 8192;
}
// padded channels of one deformable group for the plan of `g` (0 = no plan).  native_ok: the direction is tiled natively.
static int pad_group_channels(const Geom &g, bool backward, bool native_ok) {
  static const int env = getenv("MDCONV_PAD_CHANNELS") ? atoi(getenv("MDCONV_PAD_CHANNELS")) : -1;
  if (g.DG == 1) {
    if (pad_channels_preferred(g)) {
      if (g.C % 64 == 0) return g.C;
      const bool to_slab = g.nd == 3 || g.C < 16 || (g.C >= 32 && g.C < 64 && g.N >= 8192);
      return to_slab ? (g.C + 63) / 64 * 64 : (g.C + 7) / 8 * 8;   // (else only C_out is padded: the NCHW kernels need 8 | C_in)
    }
    // What the kernels do not tile at all -- C_in that is not a multiple of 8 in the backward (100 -> 100 at 40 x 40, B = 8:
    // 4.85 ms on the shape-generic kernels, 0.27 ms as 104 channels), channel counts below 16 that the size rules above leave
    // alone: the smallest padded problem, from 512 output pixels (experiment log 24).
    if (native_ok || env == 0 || g.N < 512) return 0;
    const int c8 = (g.C + 7) / 8 * 8;
    return c8 < 16 ? 16 : c8;
  }
  if (native_ok) return 0;
  const int cdp_b = g.Cdg <= 64 ? 64 : (g.Cdg <= 128 ? 128 : (g.Cdg + 255) / 256 * 256);
  const int cdp_f = (g.Cdg + 2 * kBK - 1) / (2 * kBK) * (2 * kBK);
  // (the cap looks at the backward's padding in both directions: a padded forward in front of a generic backward is no gain)
  if (dg_plan_env() != 1 && cdp_b > kPadMaxGrowth * g.Cdg) return 0;
  return backward ? cdp_b : cdp_f;
}
bool pad_plan(const Geom &g, int dtype, bool backward, PadPlan *p) {
  if (dg_plan_env() == 2) return false;
  const bool native_ok = native_supported(g, dtype, backward);
  Geom gp = g;
  if (g.G == 1) {
    const int cdp = pad_group_channels(g, backward, native_ok);
    if (cdp == 0) return false;
    // output channels below the kernels' floor of 16: padded too (with several deformable groups from 512 output pixels)
    const int Op = g.O < 16 && (g.DG == 1 || g.N >= 512) ? 16 : g.O;
    p->ng = g.DG; p->cin = g.Cdg; p->cinp = cdp;
    p->nog = 1; p->og = g.O; p->ogp = Op;
    p->wsub = g.DG;
    gp.C = gp.Cg = g.DG * cdp;
    gp.Cdg = cdp;
    gp.O = gp.Og = Op;
  } else {
    // conv groups (one deformable group): per-group channel counts the kernels do not tile -- C_in / G not a multiple of 8 or
    // below 16, fewer than 16 output channels per group -- padded PER CONV GROUP, from 512 output pixels (experiment log 28)
    static const int env = getenv("MDCONV_PAD_CHANNELS") ? atoi(getenv("MDCONV_PAD_CHANNELS")) : -1;
    p->nog = g.G; p->og = g.Og; p->ogp = g.Og < 16 ? 16 : g.Og;
    if (g.DG == 1) {
      const bool slab = pad_channels_preferred(g);   // 3-D: whole 64-channel slabs per group for the channels-last kernels
      if (env == 0 || (!slab && (native_ok || g.N < 512))) return false;
      const int c8 = (g.Cg + 7) / 8 * 8;
      p->ng = g.G; p->cin = g.Cg; p->cinp = slab ? (g.Cg + 63) / 64 * 64 : (c8 < 16 ? 16 : c8);
      p->wsub = 1;
      gp.Cg = p->cinp;
      gp.C = gp.Cdg = g.G * p->cinp;
    } else {
      // conv groups AND deformable groups the kernels do not tile, NESTED (one grouping refines the other, so that padding the
      // finer groups by the same amount keeps every channel in its conv group and its deformable group): the deformable group
      // goes to the next size the kernels tile that the finer groups divide (experiment log 29)
      if (native_ok || env == 0 || g.N < 512) return false;
      const int u = g.Cg < g.Cdg ? g.Cg : g.Cdg;   // the finer group
      if (g.Cg % u || g.Cdg % u) return false;
      const int m = g.Cdg / u;                      // finer groups per deformable group
      int cdp = 0;
      if (backward) {
        for (int cand : {64, 128, 256, 512, 768, 1024})
          if (cand >= g.Cdg && cand % m == 0) { cdp = cand; break; }
      } else {
        cdp = (g.Cdg + 2 * kBK * m - 1) / (2 * kBK * m) * (2 * kBK * m);   // finer groups of whole 32-channel stages
      }
      if (cdp == 0 || cdp > kPadMaxGrowth * g.Cdg) return false;
      p->ng = g.C / u; p->cin = u; p->cinp = cdp / m;
      p->wsub = g.Cg / u;
      gp.Cg = p->wsub * p->cinp;
      gp.Cdg = cdp;
      gp.C = p->ng * p->cinp;
    }
    gp.Og = p->ogp;
    gp.O = g.G * p->ogp;
  }
  p->pad_c = p->cinp != p->cin;
  p->pad_o = p->ogp != p->og;
  if (!p->pad_c && !p->pad_o) return false;
  if (!native_supported(gp, dtype, backward)) return false;
  p->gp = gp;
  const size_t es = dtype == MDCONV_F32 ? 4 : 2;
  size_t off = 0;
  auto take = [&](size_t &slot, size_t elems) { slot = off; off += align_up(elems * es); };
  take(p->off_x, p->pad_c ? (size_t)g.B * gp.C * g.S_i : 0);
  take(p->off_w, (size_t)gp.O * gp.Cg * g.K);
  take(p->off_gi, backward && p->pad_c ? (size_t)g.B * gp.C * g.S_i : 0);
  take(p->off_gw, backward ? (size_t)gp.O * gp.Cg * g.K : 0);
  take(p->off_o, p->pad_o ? (size_t)g.B * gp.O * g.S_o : 0);
  take(p->off_b, p->pad_o && g.with_bias && !backward ? (size_t)gp.O : 0);
  take(p->off_gb, p->pad_o && g.with_bias && backward ? (size_t)gp.O : 0);
  p->off_sub = off;
  p->total = off + native_workspace_bytes(gp, dtype, backward);
  return true;
}
// input [B][groups][cin][S_i] -> [B][groups][cinp][S_i]; weight [groups_o][og][wsub][cin][K] -> [groups_o][ogp][wsub][cinp][K] (the
// rows og .. ogp - 1 of every output group zero): rows of one (image | output channel, group), contiguous on both sides
int pad_inputs(const Geom &g, int dtype, const PadPlan &p, const Tensors &t, char *base, Tensors *tp, hipStream_t stream) {
  const size_t es = dtype == MDCONV_F32 ? 4 : 2;
  int rc;
  if (p.pad_c) {
    if ((rc = pad_rows(base + p.off_x, (size_t)p.cinp * g.S_i * es, t.input, (size_t)p.cin * g.S_i * es, (size_t)g.B * p.ng, stream)))
      return rc;
    tp->input = base + p.off_x;
  }
  if ((rc = pad_rows_grouped(base + p.off_w, (size_t)p.cinp * g.K * es, t.weight, (size_t)p.cin * g.K * es, (size_t)p.og * p.wsub,
                             (size_t)p.ogp * p.wsub, p.nog, stream)))
    return rc;
  tp->weight = base + p.off_w;
  return MDCONV_OK;
}
int pad_forward(const Geom &g, int dtype, const PadPlan &p, const Tensors &t, void *ws, hipStream_t stream) {
  char *base = (char *)ws;
  const size_t es = dtype == MDCONV_F32 ? 4 : 2;
  int rc;
  Tensors tp = t;
  if ((rc = pad_inputs(g, dtype, p, t, base, &tp, stream))) return rc;
  if (p.pad_o) {   // the kernels write ogp output channels per group (and read as many bias values): a workspace tile, real rows copied out
    if (g.with_bias) {
      if ((rc = pad_rows(base + p.off_b, (size_t)p.ogp * es, t.bias, (size_t)p.og * es, p.nog, stream))) return rc;
      tp.bias = base + p.off_b;
    }
    tp.output = base + p.off_o;
  }
  if ((rc = native_forward(p.gp, dtype, tp, base + p.off_sub, stream))) return rc;
  if (!p.pad_o) return MDCONV_OK;
  const size_t w_o = (size_t)p.og * g.S_o * es;
  return copy_rows(t.output, w_o, base + p.off_o, (size_t)p.ogp * g.S_o * es, w_o, (size_t)g.B * p.nog, stream);
}
int pad_backward(const Geom &g, int dtype, const PadPlan &p, const Tensors &t, void *ws, hipStream_t stream) {
  char *base = (char *)ws;
  const size_t es = dtype == MDCONV_F32 ? 4 : 2;
  const size_t w_x = (size_t)p.cin * g.S_i * es, p_x = (size_t)p.cinp * g.S_i * es;
  const size_t w_w = (size_t)p.cin * g.K * es, p_w = (size_t)p.cinp * g.K * es;
  const size_t w_o = (size_t)p.og * g.S_o * es, p_o = (size_t)p.ogp * g.S_o * es;
  const size_t wi = (size_t)p.og * p.wsub, wip = (size_t)p.ogp * p.wsub;   // weight rows of one output group (caller's / padded)
  int rc;
  Tensors tp = t;   // grad_offset / grad_mask have no channel axis: written in place, in the caller's mode
  if ((rc = pad_inputs(g, dtype, p, t, base, &tp, stream))) return rc;
  // accumulate modes: the padded gradient buffers start from the caller's values (like the slices above)
  if (p.pad_c) {
    if (g.acc_data && (rc = pad_rows(base + p.off_gi, p_x, t.grad_input, w_x, (size_t)g.B * p.ng, stream))) return rc;
    tp.grad_input = base + p.off_gi;
  }
  if (g.acc_w && (rc = pad_rows_grouped(base + p.off_gw, p_w, t.grad_weight, w_w, wi, wip, p.nog, stream))) return rc;
  tp.grad_weight = base + p.off_gw;
  if (p.pad_o) {
    if ((rc = pad_rows(base + p.off_o, p_o, t.grad_output, w_o, (size_t)g.B * p.nog, stream))) return rc;   // zero planes for the padding channels
    tp.grad_output = base + p.off_o;
    if (g.with_bias) {
      if (g.acc_w && (rc = pad_rows(base + p.off_gb, (size_t)p.ogp * es, t.grad_bias, (size_t)p.og * es, p.nog, stream))) return rc;
      tp.grad_bias = base + p.off_gb;
    }
  }
  if ((rc = native_backward(p.gp, dtype, tp, base + p.off_sub, stream))) return rc;
  if (p.pad_c && (rc = copy_rows(t.grad_input, w_x, base + p.off_gi, p_x, w_x, (size_t)g.B * p.ng, stream))) return rc;
  if ((rc = unpad_rows_grouped(t.grad_weight, w_w, base + p.off_gw, p_w, wi, wip, p.nog, stream))) return rc;
  if (p.pad_o && g.with_bias &&
      (rc = copy_rows(t.grad_bias, (size_t)p.og * es, base + p.off_gb, (size_t)p.ogp * es, (size_t)p.og * es, p.nog, stream)))
    return rc;
  return record_weight_ready(stream);   // after the copy back
}
}  // namespace

// native tiling unless the padded problem is the faster one (pad_channels_preferred)
static bool run_native(const Geom &g, int dtype, bool backward) {
  PadPlan pp;
  if (pad_channels_preferred(g) && pad_plan(g, dtype, backward, &pp)) return false;
  return native_supported(g, dtype, backward);
}

bool mfma_supported(const Geom &g, int dtype, bool backward) {
  if (native_supported(g, dtype, backward)) return true;
  PadPlan pp;
  if (pad_plan(g, dtype, backward, &pp)) return true;
  SplitPlan p;
  SplitFwdPlan pf;
  return backward ? split_plan(g, dtype, &p) : split_fwd_plan(g, dtype, &pf);
}

size_t mfma_workspace_bytes(const Geom &g, int dtype, bool backward) {
  if (run_native(g, dtype, backward)) return native_workspace_bytes(g, dtype, backward);
  PadPlan pp;
  if (pad_plan(g, dtype, backward, &pp)) return pp.total;
  SplitPlan p;
  SplitFwdPlan pf;
  if (backward) return split_plan(g, dtype, &p) ? p.total : 0;
  return split_fwd_plan(g, dtype, &pf) ? pf.total : 0;
}

int mfma_forward(const Geom &g, int dtype, const Tensors &t, void *ws, hipStream_t stream) {
  if (run_native(g, dtype, false)) return native_forward(g, dtype, t, ws, stream);
  PadPlan pp;
  if (pad_plan(g, dtype, false, &pp)) return pad_forward(g, dtype, pp, t, ws, stream);
  SplitFwdPlan p;
  if (!split_fwd_plan(g, dtype, &p)) { set_error("mfma_forward: no plan"); return MDCONV_EUNSUPPORTED; }
  return split_forward(g, dtype, p, t, ws, stream);
}

int mfma_backward(const Geom &g, int dtype, const Tensors &t, void *ws, hipStream_t stream) {
  if (run_native(g, dtype, true)) return native_backward(g, dtype, t, ws, stream);
  PadPlan pp;
  if (pad_plan(g, dtype, true, &pp)) return pad_backward(g, dtype, pp, t, ws, stream);
  SplitPlan p;
  if (!split_plan(g, dtype, &p)) { set_error("mfma_backward: no plan"); return MDCONV_EUNSUPPORTED; }
  return split_backward(g, dtype, p, t, ws, stream);
}

}  // namespace mdconv

extern "C" {
int mdconv_profile_enable(int on) {
  const int prev = mdconv::g_prof_on.exchange(on != 0) ? 1 : 0;
  return prev;
}
void mdconv_profile_reset(void) {
  std::lock_guard<std::mutex> lock(mdconv::g_prof_mu);
  for (int i = 0; i < mdconv::kProfSlots; ++i) mdconv::g_prof_used[i] = 0;
}
const char *mdconv_profile_name(int which) {
  if (which < 0 || which >= mdconv::kProfSlots) return "";
  std::lock_guard<std::mutex> lock(mdconv::g_prof_mu);
  return mdconv::g_prof_name[which];
}
int mdconv_profile_read(int which, double *total_ms) {
  if (which < 0 || which >= mdconv::kProfSlots) return 0;
  double tot = 0;
  int n = 0;
  std::lock_guard<std::mutex> lock(mdconv::g_prof_mu);
  for (size_t i = 0; i < mdconv::g_prof_used[which]; ++i) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, mdconv::g_prof[which][i].a, mdconv::g_prof[which][i].b) == hipSuccess) {
      tot += ms;
      ++n;
    }
  }
  if (total_ms) *total_ms = tot;
  return n;
}
}
