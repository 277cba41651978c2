// ubench_scatter2.hip -- channel-major (lanes = channels) variants: NHWC coalesced global atomics,
// NHWC coalesced gathers, conflict-free LDS ds_add_f32.  Same synthetic cfg2 sampling pattern.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int H = 56, W = 56, S = H * W, K = 9;

// block = 256 threads = 4 waves; wave handles 64 channels [lane]; block handles PT output pixels
// x all 256 channels (wave w -> channels w*64..).  NHWC buffer [B][S][C].
template <int MODE, int PT>
__global__ __launch_bounds__(256) void k_nhwc(const int *__restrict__ idx, float *__restrict__ buf, float *__restrict__ sink, int B, int C) {
  const int tiles = S / PT;
  const int b = blockIdx.x / tiles, p0 = (blockIdx.x % tiles) * PT;
  const int c = threadIdx.x;  // channel
  float acc = 0.f;
  float *img = buf + (size_t)b * S * C;
  for (int p = p0; p < p0 + PT; ++p) {
#pragma unroll
    for (int t = 0; t < K; ++t) {
      const int i0 = idx[(b * K + t) * S + p];  // uniform -> scalar load
      float *q = img + (size_t)i0 * C + c;
      if (MODE == 0) {
        acc += q[0] + q[C] + q[W * C] + q[(W + 1) * C];
      } else {
        const float v = 1.0f + t;
        unsafeAtomicAdd(q, v);
        unsafeAtomicAdd(q + C, v);
        unsafeAtomicAdd(q + W * C, v);
        unsafeAtomicAdd(q + (W + 1) * C, v);
      }
    }
  }
  if (MODE == 0) sink[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

// NCHW strided gather with lanes = channels (each lane its own plane)
__global__ __launch_bounds__(256) void k_nchw_lanec(const int *__restrict__ idx, const float *__restrict__ planes, float *__restrict__ sink, int B, int C, int PT) {
  const int tiles = S / PT;
  const int b = blockIdx.x / tiles, p0 = (blockIdx.x % tiles) * PT;
  const int c = threadIdx.x;
  const float *plane = planes + (size_t)(b * C + c) * S;
  float acc = 0.f;
  for (int p = p0; p < p0 + PT; ++p) {
#pragma unroll
    for (int t = 0; t < K; ++t) {
      const int i0 = idx[(b * K + t) * S + p];
      acc += plane[i0] + plane[i0 + 1] + plane[i0 + W] + plane[i0 + W + 1];
    }
  }
  sink[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

// LDS window [pos][64 ch] per block of 64 channels; lanes = channels -> conflict-free ds_add_f32.
// block = 256 threads: wave w handles pixels p0 + w, p0 + w + 4, ... ; all waves share the window.
template <int ROWS>
__global__ __launch_bounds__(256) void k_lds_lanec(const int *__restrict__ idx, float *__restrict__ buf, int B, int C, int PT) {
  extern __shared__ float win[];  // [ROWS*W][64]
  const int tiles = S / PT;
  const int b = blockIdx.x / tiles, p0 = (blockIdx.x % tiles) * PT;
  const int cg = blockIdx.y;  // 64-channel group
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = p0 / W - (ROWS - (PT + W - 1) / W) / 2;
  const int w0 = r0 * W;
  for (int i = threadIdx.x; i < ROWS * W * 64; i += 256) win[i] = 0.f;
  __syncthreads();
  for (int p = p0 + wave; p < p0 + PT; p += 4) {
#pragma unroll
    for (int t = 0; t < K; ++t) {
      const int i0 = idx[(b * K + t) * S + p] - w0;
      const float v = 1.0f + t;
      if (i0 >= 0 && i0 + W + 1 < ROWS * W) {
        atomicAdd(&win[i0 * 64 + lane], v);
        atomicAdd(&win[(i0 + 1) * 64 + lane], v);
        atomicAdd(&win[(i0 + W) * 64 + lane], v);
        atomicAdd(&win[(i0 + W + 1) * 64 + lane], v);
      }
    }
  }
  __syncthreads();
  // flush (NHWC scratch, coalesced atomics)
  float *img = buf + (size_t)b * S * C + cg * 64;
  for (int i = threadIdx.x; i < ROWS * W * 64; i += 256) {
    const int pos = w0 + (i >> 6);
    const float v = win[i];
    if (pos >= 0 && pos < S && v != 0.f) unsafeAtomicAdd(img + (size_t)pos * C + (i & 63), v);
  }
}

int main(int argc, char **argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32, C = 256;
  std::vector<int> h((size_t)B * K * S);
  srand(1);
  auto gauss = []() { float s = 0; for (int i = 0; i < 12; ++i) s += rand() / (float)RAND_MAX; return s - 6.f; };
  for (int b = 0; b < B; ++b)
    for (int t = 0; t < K; ++t)
      for (int p = 0; p < S; ++p) {
        int y = p / W + t / 3 - 1 + (int)floorf(gauss()), x = p % W + t % 3 - 1 + (int)floorf(gauss());
        y = y < 0 ? 0 : (y > H - 2 ? H - 2 : y);
        x = x < 0 ? 0 : (x > W - 2 ? W - 2 : x);
        h[((size_t)b * K + t) * S + p] = y * W + x;
      }
  int *d_idx; float *d_buf, *d_sink;
  CK(hipMalloc(&d_idx, h.size() * 4));
  CK(hipMemcpy(d_idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_buf, (size_t)B * C * S * 4 + (1 << 20)));
  CK(hipMemset(d_buf, 0, (size_t)B * C * S * 4 + (1 << 20)));
  CK(hipMalloc(&d_sink, (size_t)B * S * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double n_ops = (double)B * C * S * K * 4;
  auto time_it = [&](const char *name, auto launch) {
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("%-36s %8.3f ms  %7.1f Gop/s\n", name, ms, n_ops / ms * 1e-6);
  };
  time_it("nhwc gather PT=16", [&]() { hipLaunchKernelGGL((k_nhwc<0, 16>), dim3(B * S / 16), dim3(256), 0, 0, d_idx, d_buf, d_sink, B, C); });
  time_it("nhwc gather PT=56", [&]() { hipLaunchKernelGGL((k_nhwc<0, 56>), dim3(B * S / 56), dim3(256), 0, 0, d_idx, d_buf, d_sink, B, C); });
  time_it("nhwc global atomics PT=16", [&]() { hipLaunchKernelGGL((k_nhwc<1, 16>), dim3(B * S / 16), dim3(256), 0, 0, d_idx, d_buf, d_sink, B, C); });
  time_it("nhwc global atomics PT=56", [&]() { hipLaunchKernelGGL((k_nhwc<1, 56>), dim3(B * S / 56), dim3(256), 0, 0, d_idx, d_buf, d_sink, B, C); });
  time_it("nchw strided gather lanes=c PT=56", [&]() { hipLaunchKernelGGL(k_nchw_lanec, dim3(B * S / 56), dim3(256), 0, 0, d_idx, d_buf, d_sink, B, C, 56); });
  CK(hipFuncSetAttribute((const void *)k_lds_lanec<10>, hipFuncAttributeMaxDynamicSharedMemorySize, 10 * W * 64 * 4));
  time_it("lds lanes=c ROWS=10 PT=56", [&]() { hipLaunchKernelGGL((k_lds_lanec<10>), dim3(B * S / 56, 4), dim3(256), 10 * W * 64 * 4, 0, d_idx, d_buf, B, C, 56); });
  time_it("lds lanes=c ROWS=10 PT=112", [&]() { hipLaunchKernelGGL((k_lds_lanec<10>), dim3(B * S / 112, 4), dim3(256), 10 * W * 64 * 4, 0, d_idx, d_buf, B, C, 112); });
  return 0;
}
