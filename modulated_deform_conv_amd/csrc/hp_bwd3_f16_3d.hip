// hp_bwd3_f16_3d.hip -- instances of the pixel-stationary 16-bit backward kernel (hp_bwd3_kernel.hpp): F16, 3-D
#include "hp_bwd3_kernel.hpp"

namespace mdconv {

int hp_bwd3_f16_3d(const Geom &g, const HpDims &hd, const Tensors &t, const void *xt, const void *wpb, void *gcol,
                   void *colbuf, int *cnt, hipStream_t stream) {
  return g.modulated ? dispatch_bwd3<3, true, F16>(g, hd, t, xt, wpb, gcol, colbuf, cnt, stream)
                     : dispatch_bwd3<3, false, F16>(g, hd, t, xt, wpb, gcol, colbuf, cnt, stream);
}

}  // namespace mdconv
