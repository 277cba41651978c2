// ubench_valu.hip -- issue cost of the VALU instructions the 16-bit kernels interpolate with (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
template <int OP> __global__ __launch_bounds__(256) void k(float* out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  unsigned p = 0x3c003800u + threadIdx.x;
  float w = 0.5f;
  unsigned q0 = p, q1 = p + 1, q2 = p + 2, q3 = p + 3;
  for (int i = 0; i < iters; ++i) {
#define REP8(S) S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7)
    if (OP == 0) {
#define S(a) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(w), "v"(w));
      REP8(S) REP8(S)
#undef S
    } else if (OP == 1) {
#define S(a) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a) : "v"(p), "v"(w));
      REP8(S) REP8(S)
#undef S
    } else if (OP == 2) {
#define S(a) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a) : "v"(p), "v"(p));
      REP8(S) REP8(S)
#undef S
    } else if (OP == 3) {
#define S(a) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(a) : "v"(p), "v"(p));
      REP8(S) REP8(S)
#undef S
    } else if (OP == 4) {
#define S(a) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a) : "v"(p));
      REP8(S) REP8(S)
#undef S
    } else if (OP == 5) {
#define S(a) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&a) : "v"(*(double*)&a1), "v"(*(double*)&a3));
      S(a0) S(a2) S(a4) S(a6) S(a0) S(a2) S(a4) S(a6) S(a0) S(a2) S(a4) S(a6) S(a0) S(a2) S(a4) S(a6)
#undef S
    } else if (OP == 6) {
#define S(a) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(a) : "v"(w), "v"(w));
      REP8(S) REP8(S)
#undef S
    } else if (OP == 7) {
#define S(a) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "v"(w), "v"(w));
      REP8(S) REP8(S)
#undef S
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + q0 + q1 + q2 + q3;
}
template <int OP> void run(const char* name, float* d, int waves_per_simd) {
  const int iters = 4096, blocks = 256 * waves_per_simd;   // 256-thread blocks: one wave per SIMD each
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<OP><<<blocks, 256>>>(d, iters);
  hipEventRecord(a);
  k<OP><<<blocks, 256>>>(d, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double instr_per_simd = 16.0 * iters * waves_per_simd;
  printf("%-18s waves/SIMD %d: %.3f ms  %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms,
         ms * 1e-3 * 2.4e9 / instr_per_simd);
}
int main() {
  float* d; hipMalloc(&d, 256 * 256 * 8 * 4);
  for (int w : {1, 4}) {
    run<0>("v_fma_f32", d, w); run<7>("v_fmac_f32", d, w); run<1>("v_fma_mix_f32", d, w); run<2>("v_dot2c_f32_f16", d, w);
    run<3>("v_pk_fma_f16", d, w); run<4>("v_cvt_f32_f16", d, w); run<5>("v_pk_fma_f32", d, w); run<6>("v_cvt_pk_f16_f32", d, w);
  }
  return 0;
}
