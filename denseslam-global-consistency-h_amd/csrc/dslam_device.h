// dslam_device.h -- device-side data layout and per-voxel / per-pixel helpers shared by the HIP kernels.
//
// HBM layout (DESIGN.md "Data layout"):
//   hash table   HashEntry[num_buckets + num_excess]      16 B entries, ordered part then excess part
//   voxel blocks uint2[num_local_blocks * 512]            8 B voxels, block = 4 KiB, x fastest then y, z
//   free lists   int[num_local_blocks], int[num_excess]   stacks, top index in SceneCounters
// All float arithmetic follows the operation order of the reference algorithm and the
// library is compiled with -ffp-contract=off, so results are bit-identical to the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace dslam {

constexpr int kBlock = 8;
constexpr int kBlock3 = 512;
constexpr int kTileEntries = 1024;  // hash entries per workgroup in the ordered-compaction sweeps
constexpr float kFarAway = 999999.9f;
constexpr float kVeryClose = 0.05f;
constexpr int kMaxRenderingBlocks = 65536 * 4;
constexpr int kTransferBlocks = 0x1000;

struct __attribute__((aligned(16))) HashEntry {
  short pos[3];
  short pad;
  int offset;
  int ptr;
};
static_assert(sizeof(HashEntry) == 16, "ITMHashEntry layout");

struct Mat4 {  // column-major, ORUtils::Matrix4f
  float m[16];
};
struct Vec4 { float x, y, z, w; };
struct Vec3 { float x, y, z; };

// Device-resident counters of a scene (host reads them back with one small copy; kernels chain on them
// without host round trips).
struct SceneCounters {
  int last_free;        // ITMLocalVBA::lastFreeBlockId
  int last_free_ex;     // ITMVoxelBlockHash::lastFreeExcessListId
  int base_free;        // values of the two above when the current commit/realloc pass started
  int base_free_ex;
  int commit_succ_vba;  // successes counted by the commit apply kernel
  int commit_succ_ex;
  int commit_requests;
  int alloc_failures;
  int error_flags;      // bit 0: ray walked more steps than the order key can encode
  int remove_count;     // entries queued for release by decay / sliding window
  int freed_excess;     // excess slots released by the current removal pass
  int swap_count;       // entries selected by the current swap pass
  long long decayed_blocks;
  long long slid_blocks;
  int pad[2];
};

struct RenderCounters {
  int no_visible;       // ITMRenderState_VH::noVisibleEntries
  int render_tiles;     // total render tiles requested by CreateExpectedDepths
  int count_result;     // CountVisibleBlocks result
  int pad;
};

// ---- arithmetic helpers (operation order = ORUtils operators) --------------------------------------------
__device__ __forceinline__ Vec4 mul(const Mat4 &M, const Vec4 &v) {
  Vec4 r;
  r.x = M.m[0] * v.x + M.m[4] * v.y + M.m[8] * v.z + M.m[12] * v.w;
  r.y = M.m[1] * v.x + M.m[5] * v.y + M.m[9] * v.z + M.m[13] * v.w;
  r.z = M.m[2] * v.x + M.m[6] * v.y + M.m[10] * v.z + M.m[14] * v.w;
  r.w = M.m[3] * v.x + M.m[7] * v.y + M.m[11] * v.z + M.m[15] * v.w;
  return r;
}

__device__ __forceinline__ int hash_index(int bx, int by, int bz, unsigned mask) {
  return (int)((((unsigned)bx * 73856093u) ^ ((unsigned)by * 19349669u) ^ ((unsigned)bz * 83492791u)) & mask);
}

__device__ __forceinline__ HashEntry load_entry(const HashEntry *table, int idx) {
  // one 16-byte load
  const uint4 raw = *reinterpret_cast<const uint4 *>(table + idx);
  HashEntry e;
  e.pos[0] = (short)(raw.x & 0xffff);
  e.pos[1] = (short)(raw.x >> 16);
  e.pos[2] = (short)(raw.y & 0xffff);
  e.pad = 0;
  e.offset = (int)raw.z;
  e.ptr = (int)raw.w;
  return e;
}

__device__ __forceinline__ void store_entry(HashEntry *table, int idx, int px, int py, int pz, int offset, int ptr) {
  uint4 raw;
  raw.x = ((unsigned)px & 0xffffu) | ((unsigned)py << 16);
  raw.y = ((unsigned)pz & 0xffffu);
  raw.z = (unsigned)offset;
  raw.w = (unsigned)ptr;
  *reinterpret_cast<uint4 *>(table + idx) = raw;
}

// voxel packing: lo = sdf | w_depth<<16 | clr0<<24 ; hi = clr1 | clr2<<8 | w_color<<16 | pad<<24
constexpr unsigned kEmptyVoxelLo = 0x00007FFFu;
constexpr unsigned kEmptyVoxelHi = 0u;

__device__ __forceinline__ float sdf_to_float(short v) { return (float)v / 32767.0f; }
__device__ __forceinline__ short float_to_sdf(float x) { return (short)(x * 32767.0f); }

// checkPointVisibility / checkBlockVisibility
template <bool SWAPPING>
__device__ __forceinline__ void check_point_vis(bool &vis, bool &vis_enl, const Vec4 &pt, const Mat4 &M, float fx,
                                                float fy, float cx, float cy, int W, int H) {
  Vec4 b = mul(M, pt);
  if (b.z < 1e-10f) return;
  b.x = fx * b.x / b.z + cx;
  b.y = fy * b.y / b.z + cy;
  if (b.x >= 0 && b.x < W && b.y >= 0 && b.y < H) {
    vis = true;
    vis_enl = true;
  } else if (SWAPPING) {
    int lx = -W / 8, ly = W + W / 8, lz = -H / 8, lw = H + H / 8;
    if (b.x >= lx && b.x < ly && b.y >= lz && b.y < lw) vis_enl = true;
  }
}

template <bool SWAPPING>
__device__ __forceinline__ void check_block_vis(bool &vis, bool &vis_enl, int px, int py, int pz, const Mat4 &M,
                                                float fx, float fy, float cx, float cy, float voxel_size, int W,
                                                int H) {
  Vec4 pt;
  const float factor = (float)kBlock * voxel_size;
  vis = false;
  vis_enl = false;
  pt.x = (float)px * factor; pt.y = (float)py * factor; pt.z = (float)pz * factor; pt.w = 1.0f;
  check_point_vis<SWAPPING>(vis, vis_enl, pt, M, fx, fy, cx, cy, W, H); if (vis) return;  // 0 0 0
  pt.z += factor; check_point_vis<SWAPPING>(vis, vis_enl, pt, M, fx, fy, cx, cy, W, H); if (vis) return;  // 0 0 1
  pt.y += factor; check_point_vis<SWAPPING>(vis, vis_enl, pt, M, fx, fy, cx, cy, W, H); if (vis) return;  // 0 1 1
  pt.x += factor; check_point_vis<SWAPPING>(vis, vis_enl, pt, M, fx, fy, cx, cy, W, H); if (vis) return;  // 1 1 1
  pt.z -= factor; check_point_vis<SWAPPING>(vis, vis_enl, pt, M, fx, fy, cx, cy, W, H); if (vis) return;  // 1 1 0
  pt.y -= factor; check_point_vis<SWAPPING>(vis, vis_enl, pt, M, fx, fy, cx, cy, W, H); if (vis) return;  // 1 0 0
  pt.x -= factor; pt.y += factor;
  check_point_vis<SWAPPING>(vis, vis_enl, pt, M, fx, fy, cx, cy, W, H); if (vis) return;  // 0 1 0
  pt.x += factor; pt.y -= factor; pt.z += factor;
  check_point_vis<SWAPPING>(vis, vis_enl, pt, M, fx, fy, cx, cy, W, H);  // 1 0 1
}

// ---- workgroup-level ordered ranks -------------------------------------------------------------------------
// Exclusive prefix sum of one int per thread over a 256-thread workgroup (4 waves), in thread order.
// `total` receives the workgroup sum.  Uses wave64 ballot-free shuffles + one LDS hop.
__device__ __forceinline__ int wave_incl_scan(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int n = __shfl_up(v, d, 64);
    if (lane >= d) v += n;
  }
  return v;
}

template <int NWAVES>
__device__ __forceinline__ int block_excl_scan(int v, int *lds_wave_sums /* [NWAVES] */, int &total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = wave_incl_scan(v);
  if (lane == 63) lds_wave_sums[wave] = incl;
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NWAVES; w++) {
    int s = lds_wave_sums[w];
    if (w < wave) off += s;
    tot += s;
  }
  total = tot;
  __syncthreads();
  return off + incl - v;
}

}  // namespace dslam
