// mfma_kernels.hpp -- entry points of the MFMA implicit-GEMM path (see mfma_*.hip).
#pragma once
#include "mdconv_common.hpp"

namespace mdconv {

bool mfma_supported(const Geom &g, int dtype, bool backward);
size_t mfma_workspace_bytes(const Geom &g, int dtype, bool backward);
int mfma_forward(const Geom &g, int dtype, const Tensors &t, void *ws, hipStream_t stream);
int mfma_backward(const Geom &g, int dtype, const Tensors &t, void *ws, hipStream_t stream);

// 16-bit tensors on the shape-generic backward kernels: fp32 copies in the workspace, one rounding per gradient
size_t direct16_workspace_bytes(const Geom &g);
int direct16_backward(const Geom &g, int dtype, const Tensors &t, void *ws, hipStream_t stream);

// records the calling thread's "grad_weight / grad_bias are final" event on `stream`
// (include/mdconv.h: mdconv_stream_wait_weight_ready)
int record_weight_ready(hipStream_t stream);

// Two independent tails of a backward on two streams: fork_side_stream() returns a library-owned stream (own
// hardware queue) that waits for everything enqueued on `stream` so far -- or nullptr (MDCONV_BWD_FORK=0);
// join_side_stream() makes `stream` wait for it.  Captures into HIP graphs.
hipStream_t fork_side_stream(hipStream_t stream);
int join_side_stream(hipStream_t stream);

// clears `bytes` (a multiple of 2) of device memory with a kernel (graph-capture friendly)
int zero_bytes(void *p, size_t bytes, hipStream_t stream);

// benchmark hooks (include/mdconv.h: mdconv_profile_*)
void profile_mark(int which, bool begin, hipStream_t stream, const char *name = nullptr);

}  // namespace mdconv
