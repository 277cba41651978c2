// ubench_mfma2.hip -- do VALU phases of one wave overlap with MFMA phases of the other waves on
// the same SIMD?  Each wave alternates 16 x v_mfma_f32_32x32x2_f32 with NV dependent-free VALU ops,
// optionally separated by a workgroup barrier.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, bool BAR>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
  f32x16 acc[2];
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i & 7] = v[i & 7] * 1.0001f + 0.5f;
    __builtin_amdgcn_sched_barrier(0);
    if (BAR) __syncthreads();
  }
  float s = 0;
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float *d; CK(hipMalloc(&d, 256 * 8192 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char *name, auto kern, int bpc) {
    const int iters = 2000, grid = 256 * bpc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, iters); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = (double)grid * 4 * iters * 16 * 2.0 * 32 * 32 * 2;
    printf("%-28s blocks/CU=%d  %7.3f ms  %6.1f TFLOP/s (MFMA only)\n", name, bpc, ms, flop / ms * 1e-9);
  };
  for (int bpc : {1, 2, 4}) {
    run("NV=0", k<0, false>, bpc);
    run("NV=128", k<128, false>, bpc);
    run("NV=256", k<256, false>, bpc);
    run("NV=512", k<512, false>, bpc);
    run("NV=256 + barrier", k<256, true>, bpc);
  }
  return 0;
}
