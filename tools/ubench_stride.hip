// ubench_stride.hip -- do K concurrent row streams whose bases are a power of two apart collide in the memory system?
// The workspace rows of the 3-D configurations are laid out [image][tap][pixel][channels]: at cfg5 a tap is exactly 16 MiB,
// at cfg4 8 MiB, and hp_gemm2 / the partial-sums kernel / hp_bwd3 touch the K = 27 taps of one pixel range side by side.
// hipcc -O3 --offload-arch=gfx950 tools/ubench_stride.hip -o tools/ubench_stride
// grid = ranges x K workgroups (tap fastest, like hp_gemm2's units); workgroup (r, t) reads (or writes) `chunk` contiguous
// bytes at  img(r) * img_stride + t * tap_stride + (r % per_img) * chunk.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((__vector_size__(16)));
template <bool WR>
__global__ __launch_bounds__(256) void k(char* base, size_t img_stride, size_t tap_stride, size_t chunk, int per_img, int K, unsigned* out) {
  // XCD-contiguous units as in the library: block i runs on XCD i % 8; give every XCD a contiguous run
  const int nb = gridDim.x, per = (nb + 7) / 8;
  const int unit = (blockIdx.x % 8) * per + blockIdx.x / 8;
  if (unit >= nb) return;
  const int r = unit / K, t = unit - r * K;
  char* p = base + (size_t)(r / per_img) * img_stride + (size_t)t * tap_stride + (size_t)(r % per_img) * chunk;
  u32x4 acc = {0, 0, 0, 0};
  for (size_t o = threadIdx.x * 16; o < chunk; o += 256 * 16 * 4) {
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t oo = o + (size_t)u * 256 * 16;
      if (WR) { if (oo < chunk) *reinterpret_cast<u32x4*>(p + oo) = acc + (unsigned)oo; }
      else v[u] = oo < chunk ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + oo)) : acc;
    }
    if (!WR) {
#pragma unroll
      for (int u = 0; u < 4; ++u) acc ^= v[u];
    }
  }
  if (acc[0] == 0x12345678u && acc[1] == 7u) out[0] = acc[2];
}
int main() {
  const int K = 27, B = 8;
  const size_t rows = 65536, row = 256;                 // cfg5: 65536 pixels x 256 B per tap and image
  const int per_img = 5, ranges = B * per_img;          // ~ hp_gemm2's 37 ranges (here 40 = 5 per image)
  const size_t chunk = rows * row / per_img / 4096 * 4096;
  char* d; unsigned* out;
  const size_t maxpad = 1 << 20;
  const size_t total = (size_t)B * K * (rows * row + maxpad) + (64 << 20);
  hipMalloc(&d, total); hipMemset(d, 1, total); hipMalloc(&out, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int wr = 0; wr < 2; ++wr)
    for (size_t pad : {(size_t)0, (size_t)256, (size_t)4096, (size_t)8192 + 256, (size_t)65536 + 4096 + 256, (size_t)(1 << 20)}) {
      const size_t tap_stride = rows * row + pad, img_stride = K * tap_stride;
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a);
        if (wr) k<true><<<ranges * K, 256>>>(d, img_stride, tap_stride, chunk, per_img, K, out);
        else k<false><<<ranges * K, 256>>>(d, img_stride, tap_stride, chunk, per_img, K, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
      }
      printf("%s  tap stride 16 MiB + %7zu B:  %.3f ms  %.2f TB/s\n", wr ? "write" : "read ", pad, best,
             (double)ranges * K * chunk / (best * 1e-3) / 1e12);
    }
  return 0;
}
