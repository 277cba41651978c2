// ubench_pkfma.hip -- issue rate of v_pk_fma_f32 against v_fma_f32 with 8 independent accumulator chains (gfx950).
// hipcc -O3 --offload-arch=gfx950 tools/ubench_pkfma.hip -o tools/ubench_pkfma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP> __global__ __launch_bounds__(256) void k(float* out, int iters) {
  f2 a[8];
  for (int i = 0; i < 8; ++i) a[i] = f2{(float)threadIdx.x + i, (float)i};
  f2 w = f2{0.5f + threadIdx.x * 1e-6f, 0.25f}, x = f2{1.0001f, 0.9999f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) {          // 16 scalar FMAs on 16 chains (a[i].x, a[i].y)
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(w.x), "v"(x.x));
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].y) : "v"(w.x), "v"(x.y));
        } else if (OP == 1) {   // 8 packed FMAs = the same 16 FMAs
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(w), "v"(x));
        } else {                // packed, weight broadcast from the low half
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[i]) : "v"(w), "v"(x));
        }
      }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> void run(const char* name, float* d, int wps) {
  const int iters = 4096, blocks = 256 * wps;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<OP><<<blocks, 256>>>(d, iters);
  hipEventRecord(a);
  k<OP><<<blocks, 256>>>(d, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double fma_per_simd = 32.0 * iters * wps;   // wave-level scalar-FMA equivalents per SIMD
  printf("%-28s waves/SIMD %d: %.3f ms  %.2f cycles per wave-level FMA (64 lanes) per SIMD at 2.4 GHz\n", name, wps, ms,
         ms * 1e-3 * 2.4e9 / fma_per_simd);
}
int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  for (int wps : {1, 2, 4}) {
    run<0>("v_fma_f32", d, wps);
    run<1>("v_pk_fma_f32", d, wps);
    run<2>("v_pk_fma_f32 bcast weight", d, wps);
  }
  return 0;
}
