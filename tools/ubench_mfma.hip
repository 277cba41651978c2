// ubench_mfma.hip -- calibrate v_mfma_f32_32x32x2_f32 throughput on this box:
//  (a) register-only, 2 / 4 accumulators per wave, 1..4 waves per SIMD
//  (b) with one ds_read_b32 B operand per MFMA pair (the forward kernel's inner loop shape)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
  __shared__ float Bs[16 * 64];
  for (int i = threadIdx.x; i < 16 * 64; i += 256) Bs[i] = (float)i * 1e-3f;
  __syncthreads();
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      float bv = b;
      if (LDS) bv = Bs[(2 * ks + (lane >> 5)) * 64 + (lane & 31) + (it & 1) * 32];
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a + i, bv, acc[i], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float *d; CK(hipMalloc(&d, 256 * 8192 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char *name, auto kern, int nacc, int blocks_per_cu) {
    const int iters = 2000, grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, iters); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = (double)grid * 4 * iters * 8 * nacc * 2.0 * 32 * 32 * 2;
    printf("%-34s blocks/CU=%d  %7.3f ms  %6.1f TFLOP/s\n", name, blocks_per_cu, ms, flop / ms * 1e-9);
  };
  for (int bpc : {1, 2, 4}) {
    run("reg-only 2 acc", k<2, false>, 2, bpc);
    run("reg-only 4 acc", k<4, false>, 4, bpc);
    run("lds-b 2 acc", k<2, true>, 2, bpc);
    run("lds-b 4 acc", k<4, true>, 4, bpc);
  }
  return 0;
}
