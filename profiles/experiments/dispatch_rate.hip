// How fast does the part START wavefronts that have something to do?  Every lane runs a chain of `hops` dependent loads in a
// small (L2-resident) table, ~0.5-1 us per hop; the grid shape varies.  If a launch of W waves takes  t0 + W * c  whatever the
// work per wave, c is the dispatch cost per wave (or per workgroup) -- the floor of every kernel in this repo that hands each
// wave a few microseconds of latency-bound work (k_mark: 5 k waves, k_integrate: 8 k, k_render: 10 k).
//   hipcc -O3 --offload-arch=gfx950 -o dispatch_rate dispatch_rate.hip && ./dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void chase(const int *__restrict__ tab, int mask, int hops, int *out) {
  int idx = (blockIdx.x * blockDim.x + threadIdx.x) & mask;
  for (int i = 0; i < hops; i++) idx = tab[idx];
  if (idx == -7) *out = idx;
}
__global__ void chase_lds(const int *__restrict__ tab, int mask, int hops, int *out) {   // the same with 16 KiB of LDS per workgroup
  __shared__ int pad[4096];
  pad[threadIdx.x] = threadIdx.x;
  __syncthreads();
  int idx = (blockIdx.x * blockDim.x + threadIdx.x + pad[(threadIdx.x * 7) & 4095]) & mask;
  for (int i = 0; i < hops; i++) idx = tab[idx];
  if (idx == -7) *out = idx;
}
int main() {
  const int n = 1 << 16;
  std::vector<int> h(n);
  for (int i = 0; i < n; i++) h[i] = (int)(((unsigned)i * 2654435761u + 12345u) & (n - 1));
  int *tab, *out;
  CK(hipMalloc(&tab, n * 4)); CK(hipMalloc(&out, 4));
  CK(hipMemcpy(tab, h.data(), n * 4, hipMemcpyHostToDevice));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int shapes[][2] = {{300, 256}, {600, 256}, {1200, 256}, {2400, 256}, {4800, 256}, {1200, 64}, {4800, 64}, {9600, 64}, {19200, 64},
                           {300, 1024}, {600, 512}, {512, 1024}, {1024, 512}, {2048, 256}, {8192, 64}, {144, 1024}, {288, 1024}};
  for (int lds = 0; lds < 2; lds++)
    for (int hops : {1, 4, 8})
      for (auto &s : shapes) {
        float best = 1e9f;
        for (int rep = 0; rep < 12; rep++) {
          CK(hipEventRecord(a));
          if (lds) hipLaunchKernelGGL(chase_lds, dim3(s[0]), dim3(s[1]), 0, 0, tab, n - 1, hops, out);
          else hipLaunchKernelGGL(chase, dim3(s[0]), dim3(s[1]), 0, 0, tab, n - 1, hops, out);
          CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
          float ms; CK(hipEventElapsedTime(&ms, a, b));
          if (rep >= 2 && ms < best) best = ms;
        }
        printf("lds %d hops %d  grid %6d x %4d  waves %6d  %7.2f us\n", lds, hops, s[0], s[1], s[0] * s[1] / 64, best * 1e3f);
      }
  return 0;
}
