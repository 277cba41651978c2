// ubench_clock.hip -- what the matrix pipe really sustains: v_mfma_f32_32x32x2_f32 back to back on every SIMD,
// for kernels of 0.1 .. 4 ms and for a 2 s train of 1 ms kernels; the shader clock is read INSIDE the kernel
// (s_memtime = shader cycles, s_memrealtime = 100 MHz) so a power-managed clock shows up as such.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k(float *out, unsigned long long *clk, int iters) {
  f32x16 acc[2];
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0);
    }
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) {
    clk[(blockIdx.x != 0) * 2] = c1 - c0;
    clk[(blockIdx.x != 0) * 2 + 1] = w1 - w0;
  }
}

int main() {
  float *d; CK(hipMalloc(&d, 256 * 8192 * 4));
  unsigned long long *clk; CK(hipMalloc(&clk, 64)); CK(hipMemset(clk, 0, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int wall_khz = 0; CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
  int clk_khz = 0; CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
  printf("wall clock rate %d kHz, advertised shader clock %d kHz\n", wall_khz, clk_khz);
  auto run = [&](int bpc, int iters, int reps, bool print) {
    const int grid = 256 * bpc;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, clk, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[4]; CK(hipMemcpy(h, clk, 32, hipMemcpyDeviceToHost));
    const double flop = (double)reps * grid * 4 * iters * 32 * 2.0 * 32 * 32 * 2;
    const double mhz = h[1] ? (double)h[0] / (double)h[1] * (wall_khz * 1e-3) : 0;
    if (print)
      printf("blocks/CU=%d iters=%6d x%4d  %8.3f ms/launch  %6.1f TFLOP/s  in-kernel clock %.0f MHz (%.1f %% of cycles in MFMA at that clock)\n",
             bpc, iters, reps, ms / reps, flop / ms * 1e-9, mhz,
             100.0 * ((double)iters * 32 * 64) / ((double)h[0] / bpc));
    return flop / ms * 1e-9;
  };
  run(1, 1000, 1, false);
  for (int bpc : {1, 2, 4})
    for (int iters : {250 / bpc, 2500 / bpc, 10000 / bpc}) run(bpc, iters, 1, true);
  printf("train of ~1 ms kernels (4 blocks/CU), 0.25 s windows:\n");
  for (int w = 0; w < 10; ++w) run(4, 600, 250, true);
  return 0;
}
