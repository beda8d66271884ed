// micro-benchmark: 1.84 MB page-locked host -> device, as a DMA (hipMemcpyAsync + wait) and as a copy kernel that reads the host
// memory directly; wall clock per upload incl. the wait, GPU otherwise idle (the mirror's sequential pattern)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(256) void k_copy(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n16, int unroll) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride * 4) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) if (i + u * stride < n16) v[u] = src[i + u * stride];
#pragma unroll
    for (int u = 0; u < 4; u++) if (i + u * stride < n16) dst[i + u * stride] = v[u];
  }
}
int main() {
  const size_t bytes = 640 * 480 * 6;
  void *h; uint4 *d;
  CK(hipHostMalloc(&h, bytes, hipHostMallocDefault)); memset(h, 1, bytes);
  CK(hipMalloc(&d, bytes));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  auto now = [] { return std::chrono::steady_clock::now(); };
  for (int mode = 0; mode < 8; mode++) {
    const int grids[] = {0, 0, 16, 32, 64, 128, 256, 512};
    const int g = grids[mode];
    double best = 1e9, sum = 0;
    for (int it = 0; it < 60; it++) {
      auto t0 = now();
      if (mode == 0) { CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); }
      else if (mode == 1) { CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s)); CK(hipEventRecord(ev, s)); CK(hipEventSynchronize(ev)); }
      else { hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, s, d, (const uint4 *)h, bytes / 16, 4); CK(hipStreamSynchronize(s)); }
      const double us = std::chrono::duration<double, std::micro>(now() - t0).count();
      if (it >= 10) { sum += us; if (us < best) best = us; }
    }
    printf("mode %d grid %d: mean %.1f us  best %.1f us  (%.1f GB/s at the mean)\n", mode, g, sum / 50, best, bytes / (sum / 50) / 1e3);
  }
  return 0;
}
