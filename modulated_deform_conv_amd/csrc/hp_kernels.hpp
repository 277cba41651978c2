// hp_kernels.hpp -- entry points of the native 16-bit (fp16 / bf16) path (hp_*.hip).
#pragma once
#include "hp_common.hpp"

namespace mdconv {

bool hp_supported(const Geom &g, int dtype, bool backward);
size_t hp_workspace_bytes(const Geom &g, int dtype, bool backward);
// false: a forward of a few pixel tiles over many K stages, faster on the fp32 matrix kernels through fp32 copies (hp_host.hip)
bool hp_forward_preferred(const Geom &g, int dtype);
int hp_forward(const Geom &g, int dtype, const Tensors &t, void *ws, hipStream_t stream);
int hp_backward(const Geom &g, int dtype, const Tensors &t, void *ws, hipStream_t stream);

HpDims hp_dims(const Geom &g);

// hp_prep.hip
int hp_nchw_to_nhwc(const Geom &g, const HpDims &hd, const void *x, void *xt, hipStream_t stream);
int hp_pack_fwd_weights(const Geom &g, const HpDims &hd, int dtype, const void *w, void *wpf,
                        int2 *ctab, hipStream_t stream);
int hp_pack_bwd_weights(const Geom &g, const HpDims &hd, int dtype, const void *w, void *wpb,
                        int4 *btab, hipStream_t stream);
// gw32 != nullptr (calls cut into batch chunks): running fp32 sum; grad_weight is written by the last chunk
// `ranges` = pixel ranges per tap in `part` (hd.ranges_w after hp_gemm2, hd.ranges after hp_bwd)
int hp_reduce_grad_weight(const Geom &g, const HpDims &hd, int ranges, int dtype, const float *part,
                          const int4 *btab, void *grad_weight, float *gw32, bool first, bool last,
                          hipStream_t stream);
int hp_grad_bias(const Geom &g, int dtype, const void *grad_output, void *grad_bias,
                 hipStream_t stream);

// hp_fwd.hip
int hp_forward_launch(const Geom &g, const HpDims &hd, int dtype, const Tensors &t, const void *xt,
                      const void *wpf, const int2 *ctab, hipStream_t stream);

// hp_fwd2.hip: the same contraction with quad-contiguous (line-wide) gathers
int hp_forward2_launch(const Geom &g, const HpDims &hd, int dtype, const Tensors &t, const void *xt,
                       const void *wpf, const int2 *ctab, hipStream_t stream);

// hp_bwd.hip: GEMM-1 + coordinate gradients + grad_col + GEMM-2, one gather pass
int hp_backward_launch(const Geom &g, const HpDims &hd, int dtype, const Tensors &t, const void *xt,
                       const void *wpb, const int4 *btab, void *gcol, float *part, int *cnt,
                       hipStream_t stream);

// hp_bwd2.hip: the same kernel with line-wide gathers (thread roles change between phases)
size_t hp_bwd2_lds_bytes(const Geom &g, const HpDims &hd);
int hp_backward2_launch(const Geom &g, const HpDims &hd, int dtype, const Tensors &t, const void *xt,
                        const void *wpb, const int4 *btab, void *gcol, float *part, int *cnt,
                        hipStream_t stream);

// hp_bwd3.hip: pixel-stationary GEMM-1 + coordinate gradients + grad_col rows + column rows (GEMM-2 is
// hp_gemm2.hip); one conv group, 1 / 2 / 4 deformable groups, Cp a power of two
bool hp_bwd3_supported(const Geom &g, const HpDims &hd);
size_t hp_bwd3_lds_bytes(const Geom &g, const HpDims &hd);
int hp_backward3_launch(const Geom &g, const HpDims &hd, int dtype, const Tensors &t, const void *xt,
                        const void *wpb, void *gcol, void *colbuf, int *cnt, hipStream_t stream);

// hp_gemm2.hip: grad_W partials = grad_out . col^T over the column rows, dense, split over pixel ranges
size_t hp_gemm2_lds_bytes(const HpDims &hd);
int hp_gemm2_launch(const Geom &g, const HpDims &hd, int dtype, const Tensors &t, const int4 *btab,
                    const void *colbuf, float *part, hipStream_t stream);

// hp_col2im.hip: inverse scatter map (counting pass inside the fused backward kernel) + gather
int hp_csr_zero(const Geom &g, int *cnt, hipStream_t stream);
int hp_csr_build(const Geom &g, int dtype, const Tensors &t, int *cnt, int *rowptr, void *entries,
                 hipStream_t stream);
int hp_col2im(const Geom &g, const HpDims &hd, int dtype, const Tensors &t, const void *gcol,
              const int *rowptr, const void *entries, hipStream_t stream);
// two-pass gather: per-anchor partial sums (every grad_col row read once) -> stencil + transpose
size_t hp_col2im_sums_bytes(const Geom &g, const HpDims &hd, int dtype);
int hp_col2im2(const Geom &g, const HpDims &hd, int dtype, const Tensors &t, const void *gcol,
               const int *rowptr, const void *entries, void *sums, hipStream_t stream);

}  // namespace mdconv
