// placeholder until the MFMA kernels land
#include "mfma_kernels.hpp"
namespace mdconv {
bool mfma_supported(const Geom &, int, bool) { return false; }
size_t mfma_workspace_bytes(const Geom &, int, bool) { return 0; }
int mfma_forward(const Geom &, int, const Tensors &, void *, hipStream_t) { return MDCONV_EUNSUPPORTED; }
int mfma_backward(const Geom &, int, const Tensors &, void *, hipStream_t) { return MDCONV_EUNSUPPORTED; }
}  // namespace mdconv
