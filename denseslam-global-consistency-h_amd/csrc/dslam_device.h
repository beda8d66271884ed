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
constexpr int kSlabShift = 14;                 // host-store slab = 0x4000 blocks = 64 MiB of page-locked memory
constexpr int kSlabBlocks = 1 << kSlabShift;
constexpr int kMaxSlabs = 512;                 // 8.4 M blocks: more than any table has entries

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
  int next_slot;        // host store (swapping): slots handed out so far
  int swapped_in;       // blocks merged from / written to the host store by the last swap-in / swap-out
  int swapped_out;
  int pad;
};

struct RenderCounters {
  int no_visible;       // ITMRenderState_VH::noVisibleEntries
  int count_result;     // CountVisibleBlocks result
  int pad[2];
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

// native 4 x u32 vector: hipcc keeps a load of this type as ONE global_load_dwordx4, whereas HIP's uint4 struct is
// scalarised into dword + dwordx2 + dword when its fields are consumed separately (checked in the ISA)
typedef unsigned __attribute__((ext_vector_type(4))) u32x4;

__device__ __forceinline__ HashEntry load_entry(const HashEntry *table, int idx) {
  // one 16-byte load
  const u32x4 raw = *reinterpret_cast<const u32x4 *>(table + idx);
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
  u32x4 raw;
  raw.x = ((unsigned)px & 0xffffu) | ((unsigned)py << 16);
  raw.y = ((unsigned)pz & 0xffffu);
  raw.z = (unsigned)offset;
  raw.w = (unsigned)ptr;
  *reinterpret_cast<u32x4 *>(table + idx) = raw;
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

// Frustum test for a SPARSE subset of a tile's entries.  In the table sweeps only ~10 % of the entries need the
// 8-corner test, scattered over the lanes; tested in place, every wavefront would run the ~250-instruction test once
// per entry slot (4 per lane) with a handful of lanes active.  Instead the workgroup compacts its candidates into
// LDS and tests them densely (one candidate per lane), then hands every lane its own results back:
// out[k] bit 0 = visible, bit 1 = visible in the enlarged frustum (0 for non-candidates).
struct TileVisScratch {
  short4 pos[kTileEntries];
  unsigned short idx[kTileEntries];
  unsigned char res[kTileEntries];
  int n;
};

template <bool SWAPPING>
__device__ __forceinline__ void tile_block_vis(TileVisScratch &s, const bool cand[4], const short4 pos[4], const Mat4 &M,
                                               float fx, float fy, float cx, float cy, float voxel_size, int W, int H,
                                               unsigned char out[4]) {
  if (threadIdx.x == 0) s.n = 0;
  *reinterpret_cast<unsigned *>(&s.res[threadIdx.x * 4]) = 0u;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (cand[k]) {
      const int j = atomicAdd(&s.n, 1);
      s.pos[j] = pos[k];
      s.idx[j] = (unsigned short)(threadIdx.x * 4 + k);
    }
  __syncthreads();
  const int n = s.n;
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const short4 b = s.pos[j];
    bool vis, vis_enl;
    check_block_vis<SWAPPING>(vis, vis_enl, b.x, b.y, b.z, M, fx, fy, cx, cy, voxel_size, W, H);
    s.res[s.idx[j]] = (unsigned char)((vis ? 1 : 0) | (vis_enl ? 2 : 0));
  }
  __syncthreads();
  const unsigned r = *reinterpret_cast<const unsigned *>(&s.res[threadIdx.x * 4]);
  out[0] = r & 0xff; out[1] = (r >> 8) & 0xff; out[2] = (r >> 16) & 0xff; out[3] = r >> 24;
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


// ---- ordered compaction over the hash table ---------------------------------------------------------------
// Every ordered compaction is count -> scan -> apply over tiles of kTileEntries consecutive entries (one
// 256-thread workgroup per tile, 4 consecutive entries per thread), so output order is ascending entry index.

// exclusive scan over up to `n` tile counts by ONE workgroup of 1024 threads; C interleaved channels
template <int C>
__device__ void scan_tiles(const int *__restrict__ counts, int *__restrict__ offsets, int n, int totals[C]) {
  __shared__ int wave_sums[C][16];
  __shared__ int carry[C];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < C) carry[tid] = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + tid;
    int v[C], incl[C];
#pragma unroll
    for (int c = 0; c < C; c++) {
      v[c] = (i < n) ? counts[i * C + c] : 0;
      incl[c] = wave_incl_scan(v[c]);
      if (lane == 63) wave_sums[c][wave] = incl[c];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < C; c++) {
      int off = carry[c];
      for (int w = 0; w < wave; w++) off += wave_sums[c][w];
      if (i < n) offsets[i * C + c] = off + incl[c] - v[c];
    }
    __syncthreads();
    if (tid < C) {
      int t = carry[tid];
      for (int w = 0; w < 16; w++) t += wave_sums[tid][w];
      carry[tid] = t;
    }
    __syncthreads();
  }
#pragma unroll
  for (int c = 0; c < C; c++) totals[c] = carry[c];
}

// generic single-channel scan: offsets per tile, total (clipped to capacity) to *total_out
static __global__ __launch_bounds__(1024) void k_scan_count(const int *tile_counts, int *tile_offsets, int n_tiles,
                                                            int *total_out, int capacity) {
  int totals[1];
  scan_tiles<1>(tile_counts, tile_offsets, n_tiles, totals);
  if (threadIdx.x == 0) *total_out = totals[0] < capacity ? totals[0] : capacity;
}

// Sum of a strided int array over [0, n) by a 256-thread workgroup (every thread gets the result).  Lets each
// apply-workgroup derive its own exclusive tile offset from the (L2-hot, <= a few thousand) per-tile counts, which
// removes the single-workgroup scan kernel -- one ~4.5 us kernel boundary -- from every ordered compaction.
__device__ __forceinline__ int block_sum_strided(const int *__restrict__ v, int n, int stride, int *lds4) {
  int s = 0;
  for (int i = threadIdx.x; i < n; i += 256) s += v[i * stride];
  for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
  if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = s;
  __syncthreads();
  const int tot = lds4[0] + lds4[1] + lds4[2] + lds4[3];
  __syncthreads();
  return tot;
}

// The same sums in two halves, so that the loads can be issued before an unrelated dependent chain (a scan with
// barriers) and reduced after it: partial_* only loads and adds per thread, reduce_* needs the whole workgroup.
__device__ __forceinline__ int partial_sum_strided(const int *__restrict__ v, int n, int stride) {
  int s = 0;
  for (int i = threadIdx.x; i < n; i += 256) s += v[i * stride];
  return s;
}
__device__ __forceinline__ int reduce_sum(int s, int *lds4) {
  for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
  if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = s;
  __syncthreads();
  const int tot = lds4[0] + lds4[1] + lds4[2] + lds4[3];
  __syncthreads();
  return tot;
}

// both the sum over [0, n_prefix) and over [0, n_total) of a strided int array (n_prefix <= n_total), one pass
__device__ __forceinline__ void block_prefix_and_total(const int *__restrict__ v, int n_prefix, int n_total, int stride,
                                                       int *lds8, int &prefix, int &total) {
  int sp = 0, st = 0;
  for (int i = threadIdx.x; i < n_total; i += 256) {
    const int x = v[i * stride];
    st += x;
    if (i < n_prefix) sp += x;
  }
  for (int d = 32; d > 0; d >>= 1) { sp += __shfl_xor(sp, d, 64); st += __shfl_xor(st, d, 64); }
  if ((threadIdx.x & 63) == 0) { lds8[threadIdx.x >> 6] = sp; lds8[4 + (threadIdx.x >> 6)] = st; }
  __syncthreads();
  prefix = lds8[0] + lds8[1] + lds8[2] + lds8[3];
  total = lds8[4] + lds8[5] + lds8[6] + lds8[7];
  __syncthreads();
}

// ---- generic flag counting -----------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void k_flag_count(const unsigned char *__restrict__ flags, int n,
                                                    int *__restrict__ tile_counts) {
  __shared__ int red[4];
  const int t0 = blockIdx.x * kTileEntries + threadIdx.x * 4;
  int c = 0;
  if (t0 < n) {
    const uchar4 v = *reinterpret_cast<const uchar4 *>(flags + t0);
    c = (v.x > 0) + (v.y > 0) + (v.z > 0) + (v.w > 0);
  }
  int tot;
  block_excl_scan<4>(c, red, tot);
  if (threadIdx.x == 0) tile_counts[blockIdx.x] = tot;
}

// fused pass 2: like k_compact_apply, but the tile's exclusive offset is the sum of the preceding tile counts,
// computed here; the last tile also publishes the (capacity-clipped) total
static __global__ __launch_bounds__(256) void k_compact_apply_fused(const unsigned char *__restrict__ flags,
                                                                  int n_entries,
                                                                  const int *__restrict__ tile_counts, int *out,
                                                                  int capacity, int *total_out) {
  __shared__ int red[4];
  const int t0 = blockIdx.x * kTileEntries + threadIdx.x * 4;
  unsigned char f[4] = {0, 0, 0, 0};
  if (t0 < n_entries) {
    const uchar4 v = *reinterpret_cast<const uchar4 *>(flags + t0);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  }
  const int c = (f[0] > 0) + (f[1] > 0) + (f[2] > 0) + (f[3] > 0);
  int tot;
  int r = block_excl_scan<4>(c, red, tot);
  const bool last = blockIdx.x == gridDim.x - 1;
  if (tot == 0 && !last) return;
  const int offset = block_sum_strided(tile_counts, blockIdx.x, 1, red);
  if (last && threadIdx.x == 0) *total_out = (offset + tot) < capacity ? (offset + tot) : capacity;
  r += offset;
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (f[k] > 0) {
      if (r < capacity) out[r] = t0 + k;
      r++;
    }
}

// pass 2 of every ordered compaction: entry index t goes to out[rank] for flags[t] > 0, ranks ascending in t
static __global__ __launch_bounds__(256) void k_compact_apply(const unsigned char *__restrict__ flags, int n_entries,
                                                       const int *__restrict__ tile_offsets, int *__restrict__ out,
                                                       int capacity) {
  __shared__ int red[4];
  const int t0 = blockIdx.x * kTileEntries + threadIdx.x * 4;
  unsigned char f[4] = {0, 0, 0, 0};
  if (t0 < n_entries) {
    const uchar4 v = *reinterpret_cast<const uchar4 *>(flags + t0);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  }
  const int c = (f[0] > 0) + (f[1] > 0) + (f[2] > 0) + (f[3] > 0);
  int tot;
  int r = block_excl_scan<4>(c, red, tot);
  if (tot == 0) return;
  r += tile_offsets[blockIdx.x];
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (f[k] > 0) {
      if (r < capacity) out[r] = t0 + k;
      r++;
    }
}

// ---- in-launch hand-off of per-tile counts -----------------------------------------------------------------
// One 8-byte word per tile and channel: {epoch (high 32), a (bits 16..31), b (bits 0..15)}, written by ONE relaxed
// agent-scope store and polled with relaxed agent-scope loads (sc1: served past the CU's L1).  The word IS the
// payload, so no fence is involved; the epoch (one per launch, never 0) makes words of earlier launches invisible.
static __device__ __forceinline__ void publish(unsigned long long *agg, int tile, unsigned epoch, int a, int b) {
  __hip_atomic_store(&agg[tile], ((unsigned long long)epoch << 32) | ((unsigned)a << 16) | (unsigned)b, __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

// sums of both fields over the tiles [0, n) (every thread of the 256-thread workgroup gets the result).  Every thread
// has its words (up to kLookbackBatch per round) in flight TOGETHER and re-polls only those that are not there yet: a
// look-back costs about one round trip to the memory side, not one per word.
constexpr int kLookbackBatch = 8;

static __device__ __forceinline__ void lookback(const unsigned long long *agg, int n, unsigned epoch, int *lds8, int &sum_a,
                                         int &sum_b) {
  int sa = 0, sb = 0;
  for (int j0 = threadIdx.x; j0 < n; j0 += 256 * kLookbackBatch) {
    unsigned long long w[kLookbackBatch];
    unsigned pending = 0;
#pragma unroll
    for (int q = 0; q < kLookbackBatch; q++) {
      const int j = j0 + 256 * q;
      w[q] = 0;
      if (j < n) {
        w[q] = __hip_atomic_load(&agg[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pending |= 1u << q;
      }
    }
    while (true) {
#pragma unroll
      for (int q = 0; q < kLookbackBatch; q++)
        if ((pending >> q) & 1u) {
          if ((unsigned)(w[q] >> 32) == epoch) {
            sa += (int)((unsigned)w[q] >> 16);
            sb += (int)((unsigned)w[q] & 0xffffu);
            pending &= ~(1u << q);
          }
        }
      if (!pending) break;
      __builtin_amdgcn_s_sleep(8);
#pragma unroll
      for (int q = 0; q < kLookbackBatch; q++)
        if ((pending >> q) & 1u) w[q] = __hip_atomic_load(&agg[j0 + 256 * q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  for (int d = 32; d > 0; d >>= 1) { sa += __shfl_xor(sa, d, 64); sb += __shfl_xor(sb, d, 64); }
  if ((threadIdx.x & 63) == 0) { lds8[threadIdx.x >> 6] = sa; lds8[4 + (threadIdx.x >> 6)] = sb; }
  __syncthreads();
  sum_a = lds8[0] + lds8[1] + lds8[2] + lds8[3];
  sum_b = lds8[4] + lds8[5] + lds8[6] + lds8[7];
  __syncthreads();
}


// A sweep tile = kSweepTile consecutive entries handled by one 256-thread workgroup, kSweepPer consecutive entries per
// thread.  Fat tiles on purpose: the tiles of a launch wait for each other's words, and with 288 tiles instead of 1152
// a look-back is one or two loads per thread, the polling traffic is 16x smaller and the publishing stores are not
// queued behind it (with 1024-entry tiles the same kernel took 23-33 us, most of it waiting for store acknowledgements
// and for the slowest predecessor; per-tile timestamps in profiles/).
constexpr int kSweepPer = 16;
constexpr int kSweepTile = 256 * kSweepPer;


}  // namespace dslam
