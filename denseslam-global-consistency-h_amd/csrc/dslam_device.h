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
  int *err_host;        // the engine's page-locked error word (dslam_engine::err_host): report_error() stores there too
};

// A kernel reports a condition the host must hear about: sticky in the scene's counters (dslam_get_stats), and -- so that
// a caller that pipelines calls and never asks for stats still learns of it -- in the engine's page-locked error word,
// which every synchronising entry point looks at once the stream has drained (capi.hip sync_check).  Error paths only.
// 16-byte accesses with the non-temporal cache policy, for blocks nobody comes back for while they could still be cached: the
// candidates of a decay pass (aged out of every visible list by definition) and the blocks a release resets.  On the 1 GiB
// S-stress map: full decay sweep 237 -> 207 us, aged-list pass 254 -> 225, window release of every block 409 -> 382
// (two alternations on one box, harness/maint_bench.py; DSLAM_MAINT_NT=0 builds the plain form).
#ifndef DSLAM_MAINT_NT
#define DSLAM_MAINT_NT 1
#endif
typedef unsigned nt_v4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 load_nt(const uint4 *a) {
#if DSLAM_MAINT_NT
  const nt_v4 r = __builtin_nontemporal_load(reinterpret_cast<const nt_v4 *>(a));
  return make_uint4(r.x, r.y, r.z, r.w);
#else
  return *a;
#endif
}
__device__ __forceinline__ void store_nt(uint4 *a, const uint4 &v) {
#if DSLAM_MAINT_NT
  const nt_v4 r = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(r, reinterpret_cast<nt_v4 *>(a));
#else
  *a = v;
#endif
}

__device__ __forceinline__ void report_error(SceneCounters *cnt, int bits) {
  atomicOr(&cnt->error_flags, bits);
  int *h = cnt->err_host;
  if (h) __hip_atomic_store(h, bits | __hip_atomic_load(h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

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
  total = __builtin_amdgcn_readfirstlane(tot);  // (uniform, and the compiler should know: loops and barriers hang off it)
  __syncthreads();
  return off + incl - v;
}


// the same for two ints per thread at once (one pair of barriers)
template <int NWAVES>
__device__ __forceinline__ void block_excl_scan2(int a, int b, int *lds /* [2 * NWAVES] */, int &ra, int &rb, int &ta, int &tb) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int ia = a, ib = b;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int na = __shfl_up(ia, d, 64), nb = __shfl_up(ib, d, 64);
    if (lane >= d) { ia += na; ib += nb; }
  }
  if (lane == 63) { lds[wave] = ia; lds[NWAVES + wave] = ib; }
  __syncthreads();
  int oa = 0, ob = 0;
  ta = 0; tb = 0;
#pragma unroll
  for (int w = 0; w < NWAVES; w++) {
    const int sa = lds[w], sb = lds[NWAVES + w];
    if (w < wave) { oa += sa; ob += sb; }
    ta += sa; tb += sb;
  }
  __syncthreads();
  ta = __builtin_amdgcn_readfirstlane(ta);
  tb = __builtin_amdgcn_readfirstlane(tb);
  ra = oa + ia - a;
  rb = ob + ib - b;
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

// Sum of a strided int array over [0, n) by a 256-thread workgroup (every thread gets the result).  Lets each
// apply-workgroup derive its own exclusive tile offset from the (L2-hot, <= a few thousand) per-tile counts, which
// removes the single-workgroup scan kernel -- one ~4.5 us kernel boundary -- from every ordered compaction.
__device__ __forceinline__ int block_sum_strided(const int *__restrict__ v, int n, int stride, int *lds4) {
  int s = 0;
  for (int i = threadIdx.x; i < n; i += 256) s += v[i * stride];
  for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
  if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = s;
  __syncthreads();
  const int tot = __builtin_amdgcn_readfirstlane(lds4[0] + lds4[1] + lds4[2] + lds4[3]);
  __syncthreads();
  return tot;
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

// ---- in-launch hand-off of per-tile counts -----------------------------------------------------------------
// One 8-byte word per tile and channel: {epoch (high 32), a (bits 16..31), b (bits 0..15)}, written by ONE relaxed
// agent-scope store and polled with relaxed agent-scope loads (sc1: served past the CU's L1).  The word IS the
// payload, so no fence is involved; the epoch (one per launch, never 0) makes words of earlier launches invisible.
static __device__ __forceinline__ void publish(unsigned long long *agg, int tile, unsigned epoch, int a, int b) {
  __hip_atomic_store(&agg[tile], ((unsigned long long)epoch << 32) | ((unsigned)a << 16) | (unsigned)b, __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

// ---- bit-packed summaries of the hash table (round 3) --------------------------------------------------------------
// The table has 1.18 M entries of which a frame touches a few thousand.  Every pass that used to enumerate all entries
// (16 + 4 + 1 bytes each) to find them now walks bitmaps of one bit per entry (147 KB for the default table):
//   scene         alloc_bits   entry holds a resident voxel block (ptr >= 0)
//   render state  vis_bits     entriesVisibleType[entry] != 0
//   engine        q1 / q2 / mark bits (two sets, used alternately: a pass clears the set of the pass before it), retest bits
// The arrays are padded to whole units of kBitTileWords words = 32768 entries (the padding stays zero), so no kernel needs
// bounds checks; the passes cut them into their own tiles (sweep: 256 words per 256-thread workgroup, selection: 256 words per
// 1024-thread workgroup) with consecutive words on consecutive threads, so thread order = entry order and ordered ranks are
// popcounts + one block scan.
constexpr int kBitTileWords = 1024;
constexpr int kBitTileEntries = kBitTileWords * 32;
static inline __host__ __device__ int bit_tiles(int entries) { return (entries + kBitTileEntries - 1) / kBitTileEntries; }

__device__ __forceinline__ unsigned sel4(const uint4 &v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }
__device__ __forceinline__ void or4(uint4 &v, int i, unsigned m) {
  v.x |= i == 0 ? m : 0u; v.y |= i == 1 ? m : 0u; v.z |= i == 2 ? m : 0u; v.w |= i == 3 ? m : 0u;
}
__device__ __forceinline__ int popc4(const uint4 &v) { return __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w); }
__device__ __forceinline__ uint4 or4v(const uint4 &a, const uint4 &b) { return make_uint4(a.x | b.x, a.y | b.y, a.z | b.z, a.w | b.w); }
__device__ __forceinline__ uint4 andn4v(const uint4 &a, const uint4 &b) { return make_uint4(a.x & ~b.x, a.y & ~b.y, a.z & ~b.z, a.w & ~b.w); }
__device__ __forceinline__ void bit_set(unsigned *bits, int t) { atomicOr(&bits[t >> 5], 1u << (t & 31)); }
__device__ __forceinline__ void bit_clear(unsigned *bits, int t) { atomicAnd(&bits[t >> 5], ~(1u << (t & 31))); }

// Tiles are taken in the order in which workgroups START (a ticket from a device counter), not by blockIdx: a tile that
// waits for the words of the tiles in front of it then only ever waits for workgroups that are already running, whatever
// else occupies the device -- no co-residency assumption (round-2 ADVICE).  The counter only grows; the host passes the
// value it had before the launch.  Every workgroup takes exactly ONE ticket (grid = number of tiles): a loop
// "while (take_ticket() < n)" around code with barriers is what hipcc turned into nested execution-mask loops whose inner
// one re-read the same ticket (the kernel never ended); straight-line code has no such freedom.
__device__ __forceinline__ int take_ticket(unsigned *counter, unsigned base, int *lds_slot) {
  if (threadIdx.x == 0) *lds_slot = (int)(atomicAdd(counter, 1u) - base);
  __syncthreads();
  // (readfirstlane: the compiler must KNOW the tile index is wave-uniform -- a loop whose exit it believes to be divergent
  // is rebuilt with execution masks, and the barriers inside then no longer match between the waves of a workgroup)
  const int b = __builtin_amdgcn_readfirstlane(*lds_slot);
  __syncthreads();
  return b;
}

// look-back over two channels at once: sums[0..1] = fields of agg_a, sums[2..3] = fields of agg_b over the tiles [0, n).
// The spin is bounded (about a second): a word that never arrives is reported (returns false), not waited for forever.
constexpr int kSpinLimit = 1 << 20;
static __device__ __forceinline__ bool lookback2(const unsigned long long *agg_a, const unsigned long long *agg_b, int n,
                                                 unsigned epoch, int *lds16, int sums[4]) {
  int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  bool ok = true;
  for (int j = threadIdx.x; j < n; j += 256) {
    unsigned long long wa = __hip_atomic_load(&agg_a[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long wb = __hip_atomic_load(&agg_b[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while ((unsigned)(wa >> 32) != epoch || (unsigned)(wb >> 32) != epoch) {
      if (++spins > kSpinLimit) { ok = false; break; }
      __builtin_amdgcn_s_sleep(4);
      if ((unsigned)(wa >> 32) != epoch) wa = __hip_atomic_load(&agg_a[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((unsigned)(wb >> 32) != epoch) wb = __hip_atomic_load(&agg_b[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s0 += (int)((unsigned)wa >> 16); s1 += (int)((unsigned)wa & 0xffffu);
    s2 += (int)((unsigned)wb >> 16); s3 += (int)((unsigned)wb & 0xffffu);
  }
  for (int d = 32; d > 0; d >>= 1) {
    s0 += __shfl_xor(s0, d, 64); s1 += __shfl_xor(s1, d, 64); s2 += __shfl_xor(s2, d, 64); s3 += __shfl_xor(s3, d, 64);
  }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { lds16[wv] = s0; lds16[4 + wv] = s1; lds16[8 + wv] = s2; lds16[12 + wv] = s3; }
  const int all_ok = __syncthreads_and(ok ? 1 : 0);
#pragma unroll
  for (int c = 0; c < 4; c++) sums[c] = __builtin_amdgcn_readfirstlane(lds16[c * 4] + lds16[c * 4 + 1] + lds16[c * 4 + 2] + lds16[c * 4 + 3]);
  __syncthreads();
  return __builtin_amdgcn_readfirstlane(all_ok) != 0;
}

// block-wide sums of up to four ints (every thread of the 256-thread workgroup gets all of them)
__device__ __forceinline__ void block_sum4(int v[4], int *lds16) {
  for (int d = 32; d > 0; d >>= 1)
#pragma unroll
    for (int c = 0; c < 4; c++) v[c] += __shfl_xor(v[c], d, 64);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int c = 0; c < 4; c++) lds16[c * 4 + wv] = v[c];
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 4; c++) v[c] = __builtin_amdgcn_readfirstlane(lds16[c * 4] + lds16[c * 4 + 1] + lds16[c * 4 + 2] + lds16[c * 4 + 3]);
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
