// ubench_mfma3.hip -- what does the forward kernel's per-chunk skeleton cost without any global
// memory?  Per iteration: [NV VALU] [2 ds_write_b32] [barrier] [4 x (ds_read2_b32 + 4 MFMA)].
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, bool WRITE, bool BAR, int CHUNKS_PER_BAR>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
  __shared__ float Bs[2 * 16 * 32];
  const int lane = threadIdx.x & 63, kh = lane >> 5;
  for (int i = threadIdx.x; i < 2 * 16 * 32; i += 256) Bs[i] = (float)i * 1e-3f;
  __syncthreads();
  f32x16 acc[2];
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a0 = threadIdx.x * 1e-3f, a1 = 1.0f + threadIdx.x * 1e-4f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a0 + i;
  const int j = threadIdx.x & 31, ksub = threadIdx.x >> 5;
  for (int it = 0; it < iters; ++it) {
    float *Bb = Bs + (it & 1) * 16 * 32;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i & 7] = v[i & 7] * 1.0001f + 0.5f;
    if (WRITE) { Bb[(ksub * 2) * 32 + j] = v[0]; Bb[(ksub * 2 + 1) * 32 + j] = v[1]; }
    if (BAR && (it % CHUNKS_PER_BAR) == 0) __syncthreads();
    const float *Br = Bb + (lane & 31) + 4 * kh * 32;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float b = Br[(8 * q + s) * 32];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1], 0, 0, 0);
      }
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
    const int iters = 4000, grid = 256 * bpc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, iters); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = (double)grid * 4 * iters * 16 * 2.0 * 32 * 32 * 2;
    printf("%-44s blocks/CU=%d  %7.3f ms  %6.1f TFLOP/s\n", name, bpc, ms, flop / ms * 1e-9);
  };
  for (int bpc : {3, 4}) {
    run("mfma+ldsread only", k<0, false, false, 1>, bpc);
    run("+ 2 ds_write", k<0, true, false, 1>, bpc);
    run("+ barrier every chunk", k<0, true, true, 1>, bpc);
    run("+ 8 VALU", k<8, true, true, 1>, bpc);
    run("+ 16 VALU", k<16, true, true, 1>, bpc);
    run("+ 32 VALU", k<32, true, true, 1>, bpc);
    run("8 VALU, barrier every 2 chunks", k<8, true, true, 2>, bpc);
  }
  return 0;
}
