#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(256) void fill_a(uint4 *v, size_t n16) {  // current k_fill_voxels
  const uint4 e = make_uint4(0x7fff, 0, 0x7fff, 0);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) v[i] = e;
}
__global__ __launch_bounds__(256) void fill_nt(uint4 *v, size_t n16) {
  typedef unsigned __attribute__((ext_vector_type(4))) u4;
  const u4 e = {0x7fff, 0, 0x7fff, 0};
  u4 *p = reinterpret_cast<u4 *>(v);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(e, p + i);
}
template <int U>
__global__ __launch_bounds__(256) void fill_chunk(uint4 *v, size_t n16) {  // each workgroup owns contiguous chunks of U KiB x 4
  const uint4 e = make_uint4(0x7fff, 0, 0x7fff, 0);
  const size_t per_wg = (size_t)256 * U;
  for (size_t base = (size_t)blockIdx.x * per_wg; base < n16; base += (size_t)gridDim.x * per_wg)
#pragma unroll
    for (int u = 0; u < U; u++) { const size_t i = base + (size_t)u * 256 + threadIdx.x; if (i < n16) v[i] = e; }
}
__global__ __launch_bounds__(256) void rmw(uint4 *v, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { uint4 x = v[i]; x.x += 1; v[i] = x; }
}
__global__ __launch_bounds__(256) void rd(const uint4 *v, size_t n16, unsigned *out) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 x = v[i]; acc += x.x ^ x.y ^ x.z ^ x.w; }
  if (acc == 0x12345678u) *out = acc;
}
__global__ __launch_bounds__(256) void mostly_empty(const int *n, uint4 *v) {
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (wave >= *n) return;
  uint4 *blk = v + (size_t)wave * 256;  // one 4 KiB block per wave, RMW
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < 4; j++) { uint4 x = blk[j * 64 + lane]; x.x += 1; blk[j * 64 + lane] = x; }
}
// one 4 KiB block per wave, RMW; LEVELS dependent loads in front of the block address: 0 = contiguous, 1 = ptr list,
// 2 = id list -> 16-byte entry holding ptr (the integration kernel's chain, plus the count read when COUNT is set)
template <int LEVELS, bool COUNT, bool HALVES>
__global__ __launch_bounds__(256) void chain(const int *n, const int *ids, const uint4 *entries, const int *ptrs, uint4 *v) {
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  const int nv = COUNT ? *n : 7884;
  if (wave >= nv) return;
  int ptr = wave;
  if (LEVELS == 1) ptr = ptrs[wave];
  if (LEVELS == 2) ptr = (int)entries[ids[wave]].w;
  uint4 *blk = v + (size_t)ptr * 256;
  const int lane = threadIdx.x & 63;
  if (HALVES) {
#pragma unroll 1
    for (int h = 0; h < 2; h++) {
      uint4 x0 = blk[(2 * h) * 64 + lane], x1 = blk[(2 * h + 1) * 64 + lane];
      x0.x += 1; x1.x += 1;
      blk[(2 * h) * 64 + lane] = x0; blk[(2 * h + 1) * 64 + lane] = x1;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; j++) { uint4 x = blk[j * 64 + lane]; x.x += 1; blk[j * 64 + lane] = x; }
  }
}
int main() {
  const size_t bytes = (size_t)1 << 30, n16 = bytes / 16;
  uint4 *v; unsigned *o; CK(hipMalloc(&v, bytes)); CK(hipMalloc(&o, 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto run = [&](const char *name, auto launch, double nbytes) {
    for (int i = 0; i < 3; i++) launch();
    hipEventRecord(a); for (int i = 0; i < 20; i++) launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); printf("%-28s %.2f TB/s\n", name, nbytes / (ms / 20) / 1e9);
  };
  for (int g : {2048, 4096, 16384, 65536}) {
    char n[64]; snprintf(n, 64, "fill grid-stride g=%d", g); run(n, [&] { hipLaunchKernelGGL(fill_a, dim3(g), dim3(256), 0, 0, v, n16); }, bytes);
  }
  run("fill nontemporal g=4096", [&] { hipLaunchKernelGGL(fill_nt, dim3(4096), dim3(256), 0, 0, v, n16); }, bytes);
  run("fill chunk U=4 g=4096", [&] { hipLaunchKernelGGL(fill_chunk<4>, dim3(4096), dim3(256), 0, 0, v, n16); }, bytes);
  run("fill chunk U=8 g=2048", [&] { hipLaunchKernelGGL(fill_chunk<8>, dim3(2048), dim3(256), 0, 0, v, n16); }, bytes);
  run("fill chunk U=4 g=65536", [&] { hipLaunchKernelGGL(fill_chunk<4>, dim3(65536), dim3(256), 0, 0, v, n16); }, bytes);
  for (int g : {2048, 8192, 65536}) { char n[64]; snprintf(n, 64, "rmw in place g=%d", g); run(n, [&] { hipLaunchKernelGGL(rmw, dim3(g), dim3(256), 0, 0, v, n16); }, 2.0 * bytes); }
  for (int g : {2048, 8192, 65536}) { char n[64]; snprintf(n, 64, "read g=%d", g); run(n, [&] { hipLaunchKernelGGL(rd, dim3(g), dim3(256), 0, 0, v, n16, o); }, bytes); }
  int *dn; CK(hipMalloc(&dn, 4));
  for (int nv : {0, 7884, 65536, 262144}) {
    CK(hipMemcpy(dn, &nv, 4, hipMemcpyHostToDevice));
    for (int g : {2048, 16384, 65536}) {
      if ((long)g * 4 < nv) continue;
      for (int i = 0; i < 3; i++) hipLaunchKernelGGL(mostly_empty, dim3(g), dim3(256), 0, 0, dn, v);
      hipEventRecord(a); for (int i = 0; i < 20; i++) hipLaunchKernelGGL(mostly_empty, dim3(g), dim3(256), 0, 0, dn, v); hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("one block per wave, nvis=%6d, grid=%6d WGs: %.1f us per launch (%.2f TB/s)\n", nv, g, ms / 20 * 1e3, nv * 8192.0 / (ms / 20) / 1e9);
    }
  }
  {  // 7884 random blocks of the 262144 in the 1 GiB buffer
    const int nv = 7884, nb = 262144;
    int *h_ptr = new int[nv], *h_ids = new int[nv]; uint4 *h_ent = new uint4[1179648];
    unsigned r = 12345u;
    for (int i = 0; i < nv; i++) { r = r * 1664525u + 1013904223u; h_ptr[i] = (int)((r >> 8) % nb); }
    // ids ascending over a 1.18 M-entry table (as the visible list is)
    for (int i = 0; i < nv; i++) { h_ids[i] = (int)((long)i * 1179648 / nv); h_ent[h_ids[i]] = make_uint4(0, 0, 0, (unsigned)h_ptr[i]); }
    int *d_ptr, *d_ids; uint4 *d_ent;
    CK(hipMalloc(&d_ptr, nv * 4)); CK(hipMalloc(&d_ids, nv * 4)); CK(hipMalloc(&d_ent, 1179648 * 16));
    CK(hipMemcpy(d_ptr, h_ptr, nv * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_ids, h_ids, nv * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_ent, h_ent, 1179648 * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(dn, &nv, 4, hipMemcpyHostToDevice));
    auto tm = [&](const char *name, auto launch) {
      for (int i = 0; i < 3; i++) launch();
      hipEventRecord(a); for (int i = 0; i < 20; i++) launch(); hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); printf("%-58s %.1f us\n", name, ms / 20 * 1e3);
    };
    tm("7884 blocks contiguous", [&] { hipLaunchKernelGGL((chain<0, false, false>), dim3(2048), dim3(256), 0, 0, dn, d_ids, d_ent, d_ptr, v); });
    tm("7884 blocks random (ptr list)", [&] { hipLaunchKernelGGL((chain<1, false, false>), dim3(2048), dim3(256), 0, 0, dn, d_ids, d_ent, d_ptr, v); });
    tm("7884 blocks random (id list -> entry)", [&] { hipLaunchKernelGGL((chain<2, false, false>), dim3(2048), dim3(256), 0, 0, dn, d_ids, d_ent, d_ptr, v); });
    tm("  + count read first", [&] { hipLaunchKernelGGL((chain<2, true, false>), dim3(2048), dim3(256), 0, 0, dn, d_ids, d_ent, d_ptr, v); });
    tm("  + count read first + two sequential halves", [&] { hipLaunchKernelGGL((chain<2, true, true>), dim3(2048), dim3(256), 0, 0, dn, d_ids, d_ent, d_ptr, v); });
  }
  return 0;
}
