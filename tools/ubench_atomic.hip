// ubench_atomic.hip -- returning / non-returning integer atomics on scattered counters by memory scope
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int SCOPE, bool RET>
__global__ __launch_bounds__(256) void k(int* cnt, unsigned mask, int iters, int* out) {
  unsigned seed = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 99u;
  int acc = 0;
  for (int i = 0; i < iters; ++i) {
    seed = seed * 1664525u + 1013904223u;
    int* p = cnt + ((seed >> 8) & mask);
    if (RET) acc += __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, SCOPE);
    else (void)__hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, SCOPE);
  }
  if (RET && acc == 0x7fffffff) out[0] = acc;
}
template <int SCOPE, bool RET> void run(const char* name, int* cnt, unsigned mask, int* out) {
  const int iters = 64, blocks = 4096;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<SCOPE, RET><<<blocks, 256>>>(cnt, mask, iters, out);
  hipEventRecord(a);
  k<SCOPE, RET><<<blocks, 256>>>(cnt, mask, iters, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-34s %.3f ms  %.1f G atomics/s\n", name, ms, (double)blocks * 256 * iters / (ms * 1e-3) / 1e9);
}
int main() {
  int* cnt; hipMalloc(&cnt, 64 << 20); hipMemset(cnt, 0, 64 << 20);
  int* out; hipMalloc(&out, 4);
  for (unsigned words : {1u << 19, 1u << 24}) {
    printf("counters: %u (%u KB)\n", words, words / 256);
    run<__HIP_MEMORY_SCOPE_AGENT, true>("agent scope, returning", cnt, words - 1, out);
    run<__HIP_MEMORY_SCOPE_AGENT, false>("agent scope, no return", cnt, words - 1, out);
    run<__HIP_MEMORY_SCOPE_WORKGROUP, true>("workgroup scope, returning", cnt, words - 1, out);
    run<__HIP_MEMORY_SCOPE_WORKGROUP, false>("workgroup scope, no return", cnt, words - 1, out);
    run<__HIP_MEMORY_SCOPE_WAVEFRONT, true>("wavefront scope, returning", cnt, words - 1, out);
  }
  return 0;
}
