// hp_bwd3_f16_2d.hip -- instances of the pixel-stationary 16-bit backward kernel (hp_bwd3_kernel.hpp): F16, 2-D
#include "hp_bwd3_kernel.hpp"

namespace mdconv {

int hp_bwd3_f16_2d(const Geom &g, const HpDims &hd, const Tensors &t, const void *xt, const void *wpb, void *gcol,
                   void *colbuf, int *cnt, hipStream_t stream) {
  return g.modulated ? dispatch_bwd3<2, true, F16>(g, hd, t, xt, wpb, gcol, colbuf, cnt, stream)
                     : dispatch_bwd3<2, false, F16>(g, hd, t, xt, wpb, gcol, colbuf, cnt, stream);
}

}  // namespace mdconv
