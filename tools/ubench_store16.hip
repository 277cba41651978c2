// ubench_store16.hip -- cost of wave-wide row stores per CU by width / flavour / footprint (gfx950), alone and mixed
// with line-wide gathers (the hp_bwd3 mix: 8 gathers + 2 row stores per iteration).
// hipcc -O3 --offload-arch=gfx950 tools/ubench_store16.hip -o tools/ubench_store16
//   W = bytes per lane (16 / 8 / 4): a wave-store writes 64 W contiguous bytes;
//   F = 0 plain global store, 1 nontemporal, 2 raw buffer store (aux 0), 3 raw buffer store nt (aux 2)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((__vector_size__(16)));
typedef unsigned int u32x2 __attribute__((__vector_size__(8)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
template <int W, int F, int NLD>
__global__ __launch_bounds__(256) void k(char* base, const char* src, unsigned span_rows, unsigned src_rows, int iters, unsigned* out) {
  const int lane = threadIdx.x & 63;
  unsigned seed = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x / 64 * 64) * 2654435761u + 12345u);
  const rsrc_t r = make_rsrc(base, 0x7fffffffu);
  u32x4 v = {(unsigned)lane, seed, 3u, 4u};
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    u32x4 ld[NLD > 0 ? NLD : 1];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      seed = seed * 5u + 0x9E3779B9u + (seed >> 13);
      const unsigned row = (seed + (unsigned)(lane >> 3) * 3u) & (src_rows - 1);
      ld[u] = *reinterpret_cast<const u32x4*>(src + (size_t)row * 256 + (lane & 7) * 16);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      seed = seed * 5u + 0x9E3779B9u + (seed >> 13);
      const unsigned row = seed & (span_rows - 1);
      const unsigned off = row * 1024u + lane * W;
      v[0] += acc;
      if (F == 0) {
        if (W == 16) *reinterpret_cast<u32x4*>(base + off) = v;
        else if (W == 8) *reinterpret_cast<u32x2*>(base + off) = u32x2{v[0], v[1]};
        else *reinterpret_cast<unsigned*>(base + off) = v[0];
      } else if (F == 1) {
        if (W == 16) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(base + off));
        else if (W == 8) __builtin_nontemporal_store(u32x2{v[0], v[1]}, reinterpret_cast<u32x2*>(base + off));
        else __builtin_nontemporal_store(v[0], reinterpret_cast<unsigned*>(base + off));
      } else {
        if (W == 16) __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)off, 0, F == 3 ? 2 : 0);
        else if (W == 8) __builtin_amdgcn_raw_buffer_store_b64(u32x2{v[0], v[1]}, r, (int)off, 0, F == 3 ? 2 : 0);
        else __builtin_amdgcn_raw_buffer_store_b32(v[0], r, (int)off, 0, F == 3 ? 2 : 0);
      }
    }
#pragma unroll
    for (int u = 0; u < NLD; ++u) acc += ld[u][0] ^ ld[u][1] ^ ld[u][2] ^ ld[u][3];
  }
  if (acc == 0x12345678u) out[0] = acc;
}
// loads only (the same gathers, no stores)
template <int NLD>
__global__ __launch_bounds__(256) void kl(const char* src, unsigned src_rows, int iters, unsigned* out) {
  const int lane = threadIdx.x & 63;
  unsigned seed = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x / 64 * 64) * 2654435761u + 12345u);
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    u32x4 ld[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      seed = seed * 5u + 0x9E3779B9u + (seed >> 13);
      const unsigned row = (seed + (unsigned)(lane >> 3) * 3u) & (src_rows - 1);
      ld[u] = *reinterpret_cast<const u32x4*>(src + (size_t)row * 256 + (lane & 7) * 16);
    }
    seed = seed * 25u + 7u;
#pragma unroll
    for (int u = 0; u < NLD; ++u) acc += ld[u][0] ^ ld[u][1] ^ ld[u][2] ^ ld[u][3];
  }
  if (acc == 0x12345678u) out[0] = acc;
}
template <typename Fn> float timeit(Fn f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
  size_t bytes = (size_t)1 << 30;
  char* d; hipMalloc(&d, bytes + 4096); hipMemset(d, 1, bytes + 4096);
  char* s; hipMalloc(&s, (size_t)64 << 20); hipMemset(s, 1, (size_t)64 << 20);
  unsigned* out; hipMalloc(&out, 4);
  const int iters = 1024, WG = 2048;
  const double wst = (double)WG * 4 * iters * 2;   // wave-stores
  const unsigned src_rows = (2u << 20) / 256;      // gathers from 2 MB (L2-resident)
  for (size_t span : {(size_t)1 << 20, (size_t)1 << 30}) {
    const unsigned rows = (unsigned)(span / 1024);
    printf("store footprint %7zu KB\n", span >> 10);
#define RUN(W, F, NLD, name) { float t = timeit([&] { k<W, F, NLD><<<WG, 256>>>(d, s, rows, src_rows, iters, out); }); \
      printf("  %-34s %.3f ms  %6.1f cyc / wave-store / CU  %5.2f TB/s stored\n", name, t, t * 1e-3 * 2.4e9 / (wst / 256), wst * 64 * W / (t * 1e-3) / 1e12); }
    RUN(16, 0, 0, "16 B/lane plain");
    RUN(16, 1, 0, "16 B/lane nontemporal");
    RUN(16, 2, 0, "16 B/lane buffer");
    RUN(16, 3, 0, "16 B/lane buffer nt");
    RUN(8, 0, 0, "8 B/lane plain");
    RUN(8, 3, 0, "8 B/lane buffer nt");
    RUN(4, 0, 0, "4 B/lane plain");
    RUN(16, 3, 8, "16 B/lane buffer nt + 8 gathers");
    RUN(16, 0, 8, "16 B/lane plain + 8 gathers");
    RUN(8, 3, 8, "8 B/lane buffer nt + 8 gathers");
  }
  { float t = timeit([&] { kl<8><<<WG, 256>>>(s, src_rows, iters, out); });
    printf("8 gathers per iteration alone: %.3f ms  %6.1f cyc / wave-load / CU\n", t, t * 1e-3 * 2.4e9 / ((double)WG * 4 * iters * 8 / 256)); }
  return 0;
}
