// ubench_fetch.hip -- calibrates rocprofv3's FETCH_SIZE on gfx950 for 4 / 8 / 16-byte-per-lane
// loads: every kernel reads the same 1 GiB buffer exactly once (coalesced stream or a
// line-granular gather of 8-byte pairs), so the true HBM read volume is known.
//   rocprofv3 --pmc FETCH_SIZE -- ./tools/ubench_fetch
#include <hip/hip_runtime.h>
#include <stdio.h>
template <typename V> __global__ void stream_k(const V* __restrict__ p, size_t n, unsigned* out) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    V v = p[i];
    const unsigned* w = reinterpret_cast<const unsigned*>(&v);
    for (unsigned k = 0; k < sizeof(V) / 4; ++k) acc ^= w[k];
  }
  if (acc == 0x12345u) out[0] = acc;
}
// 8-byte gathers, one per 128-byte line slot in a permuted order (every line is touched 16 times
// by different waves at different times: the deformable-gather pattern of the fp32 NCHW kernels)
__global__ void gather8_k(const uint2* __restrict__ p, size_t n, unsigned* out) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t line = (i * 2654435761ull) % (n / 16), slot = (i / (n / 16)) % 16;
    uint2 v = p[line * 16 + slot];
    acc ^= v.x ^ v.y;
  }
  if (acc == 0x12345u) out[0] = acc;
}
int main() {
  const size_t bytes = (size_t)1 << 30;
  void* d; hipMalloc(&d, bytes); hipMemset(d, 1, bytes);
  unsigned* out; hipMalloc(&out, 4);
  stream_k<unsigned><<<4096, 256>>>((const unsigned*)d, bytes / 4, out);
  stream_k<uint2><<<4096, 256>>>((const uint2*)d, bytes / 8, out);
  stream_k<uint4><<<4096, 256>>>((const uint4*)d, bytes / 16, out);
  gather8_k<<<4096, 256>>>((const uint2*)d, bytes / 8, out);
  hipDeviceSynchronize();
  printf("each kernel reads %zu bytes once\n", bytes);
  return 0;
}
