// ubench_scatter.hip -- how fast are deformable-conv-shaped gathers and float atomics on gfx950?
// Emulates cfg2: B x C planes of 56x56, each lane owns an output pixel, samples 9 taps x 4 corners
// around it with N(0,1)-like jitter.  Build: hipcc --offload-arch=gfx950 -O3 ubench_scatter.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int H = 56, W = 56, S = H * W, K = 9;

// mode 0: gather (loads), 1: global atomics, 2: LDS-privatised atomics for a row-band tile
template <int MODE>
__global__ __launch_bounds__(256) void k(const int *__restrict__ idx /*[B][K][S] base index*/,
                                         float *__restrict__ planes, float *__restrict__ sink, int B, int C, int CPB) {
  // block = 256 pixels of one image; loops over CPB channels
  const int tiles = (S + 255) / 256;
  const int b = blockIdx.x / tiles;
  const int pix = (blockIdx.x % tiles) * 256 + threadIdx.x;
  const int c0 = blockIdx.y * CPB;
  if (pix >= S) return;
  int base[K];
#pragma unroll
  for (int t = 0; t < K; ++t) base[t] = idx[(b * K + t) * S + pix];
  float acc = 0.f;
  for (int c = c0; c < c0 + CPB; ++c) {
    float *plane = planes + (size_t)(b * C + c) * S;
#pragma unroll
    for (int t = 0; t < K; ++t) {
      const int i0 = base[t];
      if (MODE == 0) {
        acc += plane[i0] + plane[i0 + 1] + plane[i0 + W] + plane[i0 + W + 1];
      } else {
        const float v = 1.0f + t;
        unsafeAtomicAdd(plane + i0, v);
        unsafeAtomicAdd(plane + i0 + 1, v);
        unsafeAtomicAdd(plane + i0 + W, v);
        unsafeAtomicAdd(plane + i0 + W + 1, v);
      }
    }
  }
  if (MODE == 0) sink[blockIdx.x * 256 + threadIdx.x + (size_t)blockIdx.y * gridDim.x * 256] = acc;
}

// LDS-privatised: block = (image b, channel-chunk, row band of RB output rows); the band's scatter
// window (RB + 2*HALO rows) x W per channel lives in LDS; out-of-window hits fall back to global.
template <int RB, int HALO, int CCH>
__global__ __launch_bounds__(256) void k_lds(const int *__restrict__ idx, float *__restrict__ planes, int B, int C) {
  constexpr int WR = RB + 2 * HALO;
  __shared__ float win[CCH][WR * W];
  const int bands = (H + RB - 1) / RB;
  const int b = blockIdx.x / bands;
  const int band = blockIdx.x % bands;
  const int r0 = band * RB;
  const int w0 = (r0 - HALO) * W;  // window start (may be negative)
  const int c0 = blockIdx.y * CCH;
  for (int i = threadIdx.x; i < CCH * WR * W; i += 256) (&win[0][0])[i] = 0.f;
  __syncthreads();
  const int npix = min(RB, H - r0) * W;
  for (int p = threadIdx.x; p < npix; p += 256) {
    const int pix = r0 * W + p;
    int base[K];
#pragma unroll
    for (int t = 0; t < K; ++t) base[t] = idx[(b * K + t) * S + pix];
#pragma unroll
    for (int cc = 0; cc < CCH; ++cc) {
      float *plane = planes + (size_t)(b * C + c0 + cc) * S;
#pragma unroll
      for (int t = 0; t < K; ++t) {
        const float v = 1.0f + t;
        const int i0 = base[t] - w0;
        if (i0 >= 0 && i0 + W + 1 < WR * W) {
          atomicAdd(&win[cc][i0], v);  // ds_add_f32
          atomicAdd(&win[cc][i0 + 1], v);
          atomicAdd(&win[cc][i0 + W], v);
          atomicAdd(&win[cc][i0 + W + 1], v);
        } else {
          unsafeAtomicAdd(plane + base[t], v);
          unsafeAtomicAdd(plane + base[t] + 1, v);
          unsafeAtomicAdd(plane + base[t] + W, v);
          unsafeAtomicAdd(plane + base[t] + W + 1, v);
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < CCH * WR * W; i += 256) {
    const int cc = i / (WR * W), j = i % (WR * W);
    const int g = w0 + j;
    const float v = win[cc][j];
    if (g >= 0 && g < S && v != 0.f) unsafeAtomicAdd(planes + (size_t)(b * C + c0 + cc) * S + g, v);
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
  int *d_idx; float *d_planes, *d_sink;
  CK(hipMalloc(&d_idx, h.size() * 4));
  CK(hipMemcpy(d_idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_planes, (size_t)B * C * S * 4 + 4096));
  CK(hipMemset(d_planes, 0, (size_t)B * C * S * 4 + 4096));
  CK(hipMalloc(&d_sink, (size_t)B * 13 * 256 * 64 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double n_ops = (double)B * C * S * K * 4;
  auto time_it = [&](const char *name, auto launch) {
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("%-28s %8.3f ms  %7.1f Gop/s\n", name, ms, n_ops / ms * 1e-6);
  };
  const int tiles = (S + 255) / 256;
  for (int cpb : {4, 16, 64}) {
    char nm[64];
    snprintf(nm, 64, "gather cpb=%d", cpb);
    time_it(nm, [&]() { hipLaunchKernelGGL(k<0>, dim3(B * tiles, C / cpb), dim3(256), 0, 0, d_idx, d_planes, d_sink, B, C, cpb); });
    snprintf(nm, 64, "global atomics cpb=%d", cpb);
    time_it(nm, [&]() { hipLaunchKernelGGL(k<1>, dim3(B * tiles, C / cpb), dim3(256), 0, 0, d_idx, d_planes, d_sink, B, C, cpb); });
  }
  time_it("lds RB=4 HALO=4 CCH=8", [&]() { hipLaunchKernelGGL((k_lds<4, 4, 8>), dim3(B * 14, C / 8), dim3(256), 0, 0, d_idx, d_planes, B, C); });
  time_it("lds RB=8 HALO=4 CCH=4", [&]() { hipLaunchKernelGGL((k_lds<8, 4, 4>), dim3(B * 7, C / 4), dim3(256), 0, 0, d_idx, d_planes, B, C); });
  time_it("lds RB=8 HALO=4 CCH=8", [&]() { hipLaunchKernelGGL((k_lds<8, 4, 8>), dim3(B * 7, C / 8), dim3(256), 0, 0, d_idx, d_planes, B, C); });
  time_it("lds RB=14 HALO=4 CCH=4", [&]() { hipLaunchKernelGGL((k_lds<14, 4, 4>), dim3(B * 4, C / 4), dim3(256), 0, 0, d_idx, d_planes, B, C); });
  return 0;
}
