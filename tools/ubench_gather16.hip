// ubench_gather16.hip -- cost of 16-byte-per-lane gathers by lane->address pattern (gfx950).
// hipcc -O3 --offload-arch=gfx950 tools/ubench_gather16.hip -o tools/ubench_gather16
// Patterns (every wave-instruction moves 1 KiB; "pixel" rows are 256 B apart):
//   0: lane -> pixel l&31, 16-byte piece l>>5      (32 rows x 32 B, pieces 32 lanes apart: hp v1)
//   1: lane -> pixel l&15, piece l>>4               (16 rows x 64 B, pieces 16 lanes apart)
//   2: lane -> pixel l>>2, piece l&3                (16 rows x 64 B, pieces in adjacent lanes)
//   3: lane -> pixel l>>3, piece l&7                (8 rows x 128 B, adjacent lanes)
//   4: fully contiguous 1 KiB
// footprint: the pixel rows are drawn pseudo-randomly from `span` bytes (L1 / L2 / MALL resident).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
struct U4 { unsigned x, y, z, w; };
template <int PAT>
__global__ __launch_bounds__(256) void k(const char* base, unsigned span_rows, int iters, unsigned* out) {
  const int lane = threadIdx.x & 63;
  int prow, piece;
  if (PAT == 0) { prow = lane & 31; piece = lane >> 5; }
  else if (PAT == 1) { prow = lane & 15; piece = lane >> 4; }
  else if (PAT == 2) { prow = lane >> 2; piece = lane & 3; }
  else if (PAT == 3) { prow = lane >> 3; piece = lane & 7; }
  else { prow = 0; piece = lane; }
  unsigned seed = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x / 64 * 64) * 2654435761u + 12345u);
  unsigned acc = 0;
  for (int it = 0; it < iters; it += 8) {
    U4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      seed = seed * 5u + 0x9E3779B9u + (seed >> 13);   // cheap (shift/add) scalar-uniform walk
      // a window of neighbouring rows (like neighbouring output pixels), pseudo-random start;
      // span_rows is a power of two
      unsigned row = (seed + (unsigned)prow * 3u) & (span_rows - 1);
      if (PAT == 4) row = seed & (span_rows - 1);
      v[u] = *reinterpret_cast<const U4*>(base + (size_t)row * 256 + piece * 16);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
template <int PAT> float run(const char* d, unsigned span_rows, int iters, unsigned* out) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<PAT><<<2048, 256>>>(d, span_rows, iters, out);
  hipEventRecord(a);
  k<PAT><<<2048, 256>>>(d, span_rows, iters, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
  size_t bytes = (size_t)512 << 20;
  char* d; hipMalloc(&d, bytes + 4096); hipMemset(d, 1, bytes + 4096);
  unsigned* out; hipMalloc(&out, 4);
  const int iters = 2048;
  const double winstr = 2048.0 * 4 * iters;   // wave-instructions
  for (size_t span : {(size_t)16 << 10, (size_t)2 << 20, (size_t)64 << 20, (size_t)512 << 20}) {
    unsigned rows = (unsigned)(span / 256);
    float t[5] = {run<0>(d, rows, iters, out), run<1>(d, rows, iters, out), run<2>(d, rows, iters, out),
                  run<3>(d, rows, iters, out), run<4>(d, rows, iters, out)};
    printf("span %6zu KB:", span >> 10);
    for (int p = 0; p < 5; ++p)
      printf("  P%d %.3f ms %5.1f cyc/winstr/CU %5.2f TB/s", p, t[p], t[p] * 1e-3 * 2.4e9 / (winstr / 256), winstr * 1024 / (t[p] * 1e-3) / 1e12);
    printf("\n");
  }
  return 0;
}
