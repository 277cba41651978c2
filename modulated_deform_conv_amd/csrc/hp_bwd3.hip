// hp_bwd3.hip -- host side of the pixel-stationary backward kernel of the native 16-bit path (hp_bwd3_kernel.hpp):
// which shapes it takes, its LDS size, and the dispatch over the four translation units that instantiate it
// (hp_bwd3_{f16,bf16}_{2d,3d}.hip -- the template has 11 (lanes per pixel, channel blocks) x 4 (k-steps) x 2 variants per
// tensor type and rank; split so that they compile side by side).
#include "hp_bwd3_kernel.hpp"

namespace mdconv {

int hp_bwd3_f16_2d(const Geom &, const HpDims &, const Tensors &, const void *, const void *, void *, void *, int *, hipStream_t);
int hp_bwd3_f16_3d(const Geom &, const HpDims &, const Tensors &, const void *, const void *, void *, void *, int *, hipStream_t);
int hp_bwd3_bf16_2d(const Geom &, const HpDims &, const Tensors &, const void *, const void *, void *, void *, int *, hipStream_t);
int hp_bwd3_bf16_3d(const Geom &, const HpDims &, const Tensors &, const void *, const void *, void *, void *, int *, hipStream_t);

size_t hp_bwd3_lds_bytes(const Geom &g, const HpDims &hd) {
  const size_t region = (size_t)32 * (hd.Cp + 8) * 2 > (size_t)kB3ChunkRows * kPP * 2 ? (size_t)32 * (hd.Cp + 8) * 2
                                                                                      : (size_t)kB3ChunkRows * kPP * 2;
  const size_t slab = hd.cblks * hd.nks > kB3SlabMaxKB ? 0 : (size_t)hd.cblks * hd.nks * 1024;   // kWG: no slab in LDS
  const size_t state = (size_t)g.DG * 32 * (2 * (1 << g.nd) + 4) * 4;                            // St[group][pixel][SW]
  return slab + 4 * (region + state);
}

bool hp_bwd3_supported(const Geom &g, const HpDims &hd) {
  if (g.G != 1) return false;
  if (g.DG != 1 && g.DG != 2 && g.DG != 4) return false;
  if (hd.Cp != 32 && hd.Cp != 64 && hd.Cp != 128 && hd.Cp != 256) return false;
  // deformable groups: whole lanes of 8 channels, no padded channels (a group's lanes tile the Cp-wide rows)
  if (g.DG > 1 && (hd.Cp != g.C || g.Cdg % 16 != 0)) return false;
  if (hd.nks > 16) return false;
  // two workgroups per CU where one group and a staged slab would otherwise take hp_bwd2 (which holds such shapes without
  // spilling); instances that have no alternative (A fragments from global memory, deformable groups) may take the CU alone
  const bool alone_ok = g.DG > 1 || hd.cblks * hd.nks > kB3SlabMaxKB;
  return hp_bwd3_lds_bytes(g, hd) <= (size_t)(alone_ok ? 160 : 80) * 1024;
}

int hp_backward3_launch(const Geom &g, const HpDims &hd, int dtype, const Tensors &t, const void *xt,
                        const void *wpb, void *gcol, void *colbuf, int *cnt, hipStream_t stream) {
  if (dtype == MDCONV_F16)
    return g.nd == 2 ? hp_bwd3_f16_2d(g, hd, t, xt, wpb, gcol, colbuf, cnt, stream)
                     : hp_bwd3_f16_3d(g, hd, t, xt, wpb, gcol, colbuf, cnt, stream);
  return g.nd == 2 ? hp_bwd3_bf16_2d(g, hd, t, xt, wpb, gcol, colbuf, cnt, stream)
                   : hp_bwd3_bf16_3d(g, hd, t, xt, wpb, gcol, colbuf, cnt, stream);
}

}  // namespace mdconv
