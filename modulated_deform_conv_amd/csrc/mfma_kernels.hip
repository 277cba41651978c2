// mfma_kernels.hip -- dispatch, workspace layout and weight packing for the MFMA path.
#include "mfma_kernels.hpp"
#include "mfma_tile.hpp"

namespace mdconv {

namespace {

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

int pack_weights_f32(const Geom &g, const PackDims &pd, const float *weight, float *wp, float *wq,
                     hipStream_t stream) {
  const int64_t total = (int64_t)g.G * g.K * pd.Cgp * pd.Ogp;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, stream, g, pd, weight, wp, wq);
  return check_launch("pack_weights");
}

bool mfma_supported(const Geom &g, int dtype, bool backward) {
  if (dtype != MDCONV_F32) return false;
  if (g.Cg < 16 || g.Og < 16) return false;  // MFMA tiles would be mostly padding
  if (!(g.DG == 1 || (g.Cdg % kBK == 0 && g.Cg % kBK == 0))) return false;
  if (backward) return false;                // backward kernels land next
  return true;
}

size_t mfma_workspace_bytes(const Geom &g, int dtype, bool backward) {
  (void)dtype;
  const PackDims pd = pack_dims(g);
  const size_t wbytes = align_up((size_t)g.G * g.K * pd.Cgp * pd.Ogp * sizeof(float));
  return backward ? 2 * wbytes : wbytes;
}

int mfma_forward(const Geom &g, int dtype, const Tensors &t, void *ws, hipStream_t stream) {
  (void)dtype;
  const PackDims pd = pack_dims(g);
  float *wp = (float *)ws;
  int rc = pack_weights_f32(g, pd, (const float *)t.weight, wp, nullptr, stream);
  if (rc) return rc;
  return mfma_forward_f32(g, pd, t, wp, stream);
}

int mfma_backward(const Geom &, int, const Tensors &, void *, hipStream_t) {
  set_error("mfma backward not implemented");
  return MDCONV_EUNSUPPORTED;
}

}  // namespace mdconv
