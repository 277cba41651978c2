// mfma_kernels.hip -- dispatch, workspace layout and weight packing for the MFMA path.
#include "mfma_kernels.hpp"
#include "mfma_tile.hpp"

#include <vector>

namespace mdconv {

namespace {

bool g_prof_on = false;
struct ProfPair { hipEvent_t a, b; };
std::vector<ProfPair> g_prof[3];
size_t g_prof_used[3] = {0, 0, 0};

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

}  // namespace

void profile_mark(int which, bool begin, hipStream_t stream) {
  if (!g_prof_on || which < 0 || which > 2) return;
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
  bd.OgpB = (g.O + 255) / 256 * 256;
  bd.mblks = bd.OgpB / 32;
  bd.mtiles = bd.OgpB / 256;
  bd.Cp = (g.C + 31) / 32 * 32;
  bd.cblks = bd.Cp / 32;
  const int col_tiles = bd.mtiles * g.K * bd.cblks;
  const int pairs = bd.Np / 32;
  int splits = (1024 + col_tiles - 1) / col_tiles;       // ~4 workgroups per CU
  if (splits > pairs) splits = pairs;
  if (splits < 1) splits = 1;
  bd.pairs_per_split = (pairs + splits - 1) / splits;
  bd.splits = (pairs + bd.pairs_per_split - 1) / bd.pairs_per_split;
  bd.ochunks = (g.O + 31) / 32 * 2;
  bd.waves_c = g.C > 128 ? 4 : (g.C > 64 ? 2 : 1);
  bd.cblks_q = (g.C + 64 * bd.waves_c - 1) / (64 * bd.waves_c) * (2 * bd.waves_c);
  const int nc = 1 << g.nd;
  size_t off = 0;
  bd.off_wq = off;   off += align_up((size_t)g.K * bd.ochunks * bd.cblks_q * 2 * 64 * 16);
  bd.off_ga = off;   off += align_up((size_t)bd.Np * bd.OgpB * sizeof(float));
  bd.off_table = off; off += align_up((size_t)g.DG * g.K * bd.Np * 2 * (1 << g.nd) * sizeof(int));
  bd.off_part = off; off += align_up((size_t)bd.splits * g.K * bd.OgpB * bd.Cp * sizeof(float));
  bd.off_gcol = off; off += align_up((size_t)g.B * g.C * g.K * g.S_o * sizeof(float));
  bd.off_cnt = off;  off += align_up((size_t)g.B * g.K * g.S_i * sizeof(int));
  bd.off_rowptr = off; off += align_up((size_t)g.B * g.K * (g.S_i + 1) * sizeof(int));
  bd.off_entries = off; off += align_up((size_t)g.B * g.K * g.S_o * nc * 8);
  bd.off_end = off;
  return bd;
}

bool mfma_supported(const Geom &g, int dtype, bool backward) {
  if (dtype != MDCONV_F32) return false;
  if (g.Cg < 16 || g.Og < 16) return false;  // MFMA tiles would be mostly padding
  if (g.in_sz[g.nd - 1] < 2) return false;   // paired-corner gathers need 2 columns
  if (!(g.DG == 1 || (g.Cdg % (2 * kBK) == 0 && g.Cg % (2 * kBK) == 0))) return false;
  // raw buffer addressing: every tensor must stay below 2 GiB
  if ((size_t)g.B * g.C * g.S_i * sizeof(float) >= ((size_t)1 << 31)) return false;
  if (backward) {
    if (g.G != 1 || g.DG != 1 || g.C % 8) return false;
    if ((size_t)g.B * g.O * g.S_o * sizeof(float) >= ((size_t)1 << 31)) return false;
    if ((size_t)g.B * g.C * g.K * g.S_o * sizeof(float) >= ((size_t)1 << 31)) return false;  // grad_col
  }
  return true;
}

size_t mfma_workspace_bytes(const Geom &g, int dtype, bool backward) {
  (void)dtype;
  if (backward) return bwd_dims(g).off_end;
  const PackDims pd = pack_dims(g);
  return align_up((size_t)g.G * g.K * pd.Cgp * pd.Ogp * sizeof(float));
}

int mfma_forward(const Geom &g, int dtype, const Tensors &t, void *ws, hipStream_t stream) {
  (void)dtype;
  const PackDims pd = pack_dims(g);
  float *wp = (float *)ws;
  int rc = pack_weights_f32(g, pd, (const float *)t.weight, wp, nullptr, stream);
  if (rc) return rc;
  profile_mark(0, true, stream);
  rc = mfma_forward_f32(g, pd, t, wp, stream);
  profile_mark(0, false, stream);
  return rc;
}

int mfma_backward(const Geom &g, int dtype, const Tensors &t, void *ws, hipStream_t stream) {
  const BwdDims bd = bwd_dims(g);
  char *base = (char *)ws;
  float *ga = (float *)(base + bd.off_ga);
  int *table = (int *)(base + bd.off_table);
  float *part = (float *)(base + bd.off_part);
  float *wq = (float *)(base + bd.off_wq);
  float *gcol = (float *)(base + bd.off_gcol);
  int rc;
  (void)dtype;
  // grad_offset / grad_mask (+ grad_col), then grad_input through the inverted scatter map
  if ((rc = pack_wq_f32(g, bd, (const float *)t.weight, wq, stream))) return rc;
  profile_mark(1, true, stream);
  rc = mfma_bwd_data_f32(g, bd, t, wq, gcol, stream);
  profile_mark(1, false, stream);
  if (rc) return rc;
  if ((rc = col2im_f32(g, bd, t, gcol, (int *)(base + bd.off_cnt), (int *)(base + bd.off_rowptr),
                       base + bd.off_entries, stream)))
    return rc;
  // grad_weight / grad_bias
  if ((rc = build_tap_table_f32(g, bd, t, table, stream))) return rc;
  if ((rc = pack_gout_f32(g, bd, (const float *)t.grad_output, ga, stream))) return rc;
  return mfma_bwd_weight_f32(g, bd, t, ga, table, part, stream);
}


}  // namespace mdconv

extern "C" {
int mdconv_profile_enable(int on) {
  const int prev = mdconv::g_prof_on ? 1 : 0;
  mdconv::g_prof_on = on != 0;
  return prev;
}
void mdconv_profile_reset(void) {
  for (int i = 0; i < 3; ++i) mdconv::g_prof_used[i] = 0;
}
int mdconv_profile_read(int which, double *total_ms) {
  if (which < 0 || which > 2) return 0;
  double tot = 0;
  int n = 0;
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
