// alloc.hip -- scene reset, view conversion and AllocateSceneFromDepth for gfx950.
//
// Reference call sites: denseMapper->ResetScene (InfiniTamDriver.h:354-360), viewBuilder->UpdateView
// (InfiniTamDriver.cpp:280-288), denseMapper->ProcessFrame -> AllocateSceneFromDepth (InfiniTamDriver.h:187-192).
// Algorithm: SURVEY.md Appendix A.3, A.4, A.10.
//
// Allocation is bit-exact with the sequential CPU engine ("last writer in row-major pixel order wins", pool
// slots handed out in ascending hash-index order) although it runs wave-parallel, in two launches:
//   k_mark         per pixel, walk the +-mu segment in block units; misses do atomicMax(order_key[slot], pixel*cap+step+1)
//   k_alloc_sweep  one pass over the table: the final key of a slot names its winner, whose walk is replayed to the
//                  block it asked for; the r-th requesting entry in hash-index order gets voxelAllocationList[lastFree - r]
//                  (pool exhaustion follows the closed form derived in DESIGN.md); entries visible in the previous pass
//                  are re-tested against the frustum; visibleEntryIDs is written ascending in hash index.  The ordered
//                  ranks come from per-tile counts exchanged INSIDE the launch (see below).
// No host round trip between the phases: every count lives in device memory (SceneCounters/RenderCounters).
#include <cstdio>
#include <cstdlib>

#include "dslam_internal.h"

#pragma clang fp contract(off)

namespace dslam {

// ---------------------------------------------------------------------------------------------------------
// ResetScene
// ---------------------------------------------------------------------------------------------------------
// One short-lived workgroup per 16 KiB (4 x 16 bytes per lane), not a persistent grid-stride loop: on MI355X a 1 GiB
// fill runs at 6.0 TB/s this way against 3.9 TB/s for 4096 looping workgroups (scratch microbenchmark).
constexpr int kFillUnroll = 4;
__global__ __launch_bounds__(256) void k_fill_voxels(uint4 *__restrict__ v, size_t n16) {
  const uint4 empty2 = make_uint4(kEmptyVoxelLo, kEmptyVoxelHi, kEmptyVoxelLo, kEmptyVoxelHi);
  const size_t base = (size_t)blockIdx.x * (256 * kFillUnroll) + threadIdx.x;
#pragma unroll
  for (int u = 0; u < kFillUnroll; u++) {
    const size_t i = base + (size_t)u * 256;
    if (i < n16) v[i] = empty2;
  }
}

__global__ __launch_bounds__(256) void k_reset_tables(HashEntry *hash, int n_entries, int *alloc_list, int *last_seen,
                                                      int n_local, int *excess_list, int n_excess,
                                                      SceneCounters *cnt) {
  const int stride = gridDim.x * blockDim.x;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = tid; i < n_entries; i += stride) store_entry(hash, i, 0, 0, 0, 0, -2);
  for (int i = tid; i < n_local; i += stride) { alloc_list[i] = i; last_seen[i] = -1; }
  for (int i = tid; i < n_excess; i += stride) excess_list[i] = i;
  if (tid == 0) {
    SceneCounters c = {};
    c.last_free = n_local - 1;
    c.last_free_ex = n_excess - 1;
    *cnt = c;
  }
}

int launch_scene_reset(dslam_engine *e, dslam_scene *s) {
  const size_t n16 = (size_t)s->p.num_local_blocks * kBlock3 / 2;
  const unsigned fill_wgs = (unsigned)((n16 + 256 * kFillUnroll - 1) / (256 * kFillUnroll));
  hipLaunchKernelGGL(k_fill_voxels, dim3(fill_wgs), dim3(256), 0, e->stream, reinterpret_cast<uint4 *>(s->voxels), n16);
  hipLaunchKernelGGL(k_reset_tables, dim3(2048), dim3(256), 0, e->stream, s->hash, s->n_entries, s->alloc_list,
                     s->last_seen, s->p.num_local_blocks, s->excess_list, s->p.num_excess, s->counters);
  DSLAM_HIP(hipMemsetAsync(s->masks, 0, (size_t)s->p.num_local_blocks * 2 * s->history_words * sizeof(unsigned long long),
                           e->stream));
  if (s->swap_state) DSLAM_HIP(hipMemsetAsync(s->swap_state, 0, s->n_entries, e->stream));
  DSLAM_HIP(hipGetLastError());
  return DSLAM_OK;
}

// ---------------------------------------------------------------------------------------------------------
// view conversion: short millimetres -> float metres, rgba copied as is
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_convert_depth(const short *__restrict__ in, float *__restrict__ out, int n,
                                                       float a, float b) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int d = in[i];
    out[i] = (d <= 0 || d > 32000) ? -1.0f : (float)d * a + b;
  }
}

// UpdateView on resident data costs nothing here: the view records where the frame lives; the float depth image is
// derived by the next consumer (the allocation pass folds it into its preparation kernel).
int launch_view_convert(dslam_engine *, dslam_view *v, const void *rgba_dev, const void *depth_dev, float a, float b) {
  v->rgba_src = reinterpret_cast<const uchar4 *>(rgba_dev);
  v->raw_src = reinterpret_cast<const short *>(depth_dev);
  v->affine_a = a; v->affine_b = b;
  v->depth_dirty = true;
  return DSLAM_OK;
}

int ensure_view_depth(dslam_engine *e, const dslam_view *v) {
  if (!v->depth_dirty) return DSLAM_OK;
  const int n = v->w_d * v->h_d;
  hipLaunchKernelGGL(k_convert_depth, dim3((n + 255) / 256), dim3(256), 0, e->stream, v->raw_src, v->depth, n,
                     v->affine_a, v->affine_b);
  DSLAM_HIP(hipGetLastError());
  v->depth_dirty = false;
  return DSLAM_OK;
}

// ---------------------------------------------------------------------------------------------------------
// AllocateSceneFromDepth
// ---------------------------------------------------------------------------------------------------------
// TWO launches (round 1 needed six, each ~4.5 us even when nearly empty on this 8-XCD part):
//   k_mark         per pixel: derive the float depth, walk the +-mu segment, mark found entries visible, race for the
//                  order key of every missing block (atomicMax: the last pixel/step in row-major order wins)
//   k_alloc_sweep  ONE pass over the table that does what used to be winners -> commit -> visible count -> compaction:
//                  an ordered compaction normally needs a grid-wide dependency between counting and placing; here every
//                  tile publishes its counts in an 8-byte {epoch, counts} word the moment it has them and the tiles
//                  behind it add up the words in front of them inside the same launch (decoupled look-back on
//                  agent-scope relaxed atomics; all workgroups are co-resident, tiles are taken in ascending order, so
//                  a tile only ever waits for tiles that are running or done).
// What made the other launches disappear:
//   * no clearing pass: the sweep zeroes exactly the keys that were set, and the mark kernel clears the allocType bytes
//     of the previous pass through the list of requests the sweep left behind;
//   * no second walk: the winner of a slot is its final key, and (pixel, step) is all it takes to replay that pixel's
//     walk to the block it asked for (same float operations, same order -> same coordinates); whether it is an
//     ordered (1) or excess (2) request follows from the entry the key sits on;
//   * no re-arming pass: the visible types carry a generation bit, so "visible in the previous pass" (upstream sets
//     those to 3 from the previous list) is simply a non-zero byte with the other bit.
struct MarkParams {
  const short *raw;    // non-null: the float depth image is derived here (UpdateView's conversion)
  float *depth;
  float a, b;
  int W, H;
  Mat4 invM;
  float inv_fx, inv_fy, cx, cy;
  float mu, frustum_min, frustum_max, one_over_block;
  const HashEntry *hash;
  unsigned mask;
  int num_buckets;
  unsigned *keys;
  unsigned char *vis_type;
  unsigned gen;        // this pass' generation bit (0 or 0x80)
  int step_cap;
  SceneCounters *cnt;
  unsigned char *alloc_type;   // allocType bytes of the previous pass are cleared through its request list
  const int *req_list;
  const int *req_count;
};

// the +-mu segment of a pixel in block units: start point, step vector, number of steps (buildHashAllocAndVisibleTypePP)
__device__ __forceinline__ int ray_segment(float d, int x, int y, const Mat4 &invM, float inv_fx, float inv_fy, float cx,
                                           float cy, float mu, float one_over_block, Vec3 &pt, Vec3 &dir) {
  Vec3 pc;
  pc.z = d;
  pc.x = pc.z * (((float)x - cx) * inv_fx);
  pc.y = pc.z * (((float)y - cy) * inv_fy);
  float norm = sqrtf(pc.x * pc.x + pc.y * pc.y + pc.z * pc.z);
  Vec4 tmp;
  tmp.x = pc.x * (1.0f - mu / norm); tmp.y = pc.y * (1.0f - mu / norm); tmp.z = pc.z * (1.0f - mu / norm); tmp.w = 1.0f;
  Vec4 q = mul(invM, tmp);
  pt.x = q.x * one_over_block; pt.y = q.y * one_over_block; pt.z = q.z * one_over_block;
  tmp.x = pc.x * (1.0f + mu / norm); tmp.y = pc.y * (1.0f + mu / norm); tmp.z = pc.z * (1.0f + mu / norm);
  q = mul(invM, tmp);
  const Vec3 pe = {q.x * one_over_block, q.y * one_over_block, q.z * one_over_block};
  dir.x = pe.x - pt.x; dir.y = pe.y - pt.y; dir.z = pe.z - pt.z;
  norm = sqrtf(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
  const int no_steps = (int)ceilf(2.0f * norm);
  const float div = (float)(no_steps - 1);
  dir.x /= div; dir.y /= div; dir.z /= div;
  return no_steps;
}

__global__ __launch_bounds__(256) void k_mark(MarkParams p) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  {  // (independent job) forget the allocType bytes of the previous pass
    const int n = *p.req_count;
    for (int i = idx; i < n; i += gridDim.x * 256) p.alloc_type[p.req_list[i]] = 0;
  }
  // every lane stays in the kernel (invalid pixels march zero steps): the order-key atomics below are aggregated
  // per wavefront, which needs the wave converged
  const bool in_image = idx < p.W * p.H;
  const int y = in_image ? idx / p.W : 0, x = in_image ? idx - y * p.W : 0;
  float d = -1.0f;
  if (in_image) {
    if (p.raw) {
      const int r = p.raw[idx];
      d = (r <= 0 || r > 32000) ? -1.0f : (float)r * p.a + p.b;
      p.depth[idx] = d;
    } else {
      d = p.depth[idx];
    }
  }
  const bool valid = !(d <= 0 || (d - p.mu) < 0 || (d - p.mu) < p.frustum_min || (d + p.mu) > p.frustum_max);
  Vec3 pt, dir;
  int no_steps = ray_segment(d, x, y, p.invM, p.inv_fx, p.inv_fy, p.cx, p.cy, p.mu, p.one_over_block, pt, dir);
  if (!valid) no_steps = 0;
  if (no_steps > p.step_cap) {  // the order key cannot encode later steps: report instead of mis-ordering
    atomicOr(&p.cnt->error_flags, 1);
    no_steps = p.step_cap;
  }
  int wave_steps = no_steps;
  for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(wave_steps, o, 64); wave_steps = v > wave_steps ? v : wave_steps; }

  // The bucket heads of the first kPre steps are requested together, before any of them is looked at: a pixel's walk is
  // 2-3 steps of one dependent 16-byte read each, and the positions of all of them are known up front (the same
  // additions in the same order as the walk below).  A lane with fewer steps reads a bucket it will not use.
  constexpr int kPre = 3;
  HashEntry pre[kPre];
  {
    Vec3 q = pt;
#pragma unroll
    for (int i = 0; i < kPre; i++) {
      pre[i] = load_entry(p.hash, hash_index((short)(int)floorf(q.x), (short)(int)floorf(q.y), (short)(int)floorf(q.z), p.mask));
      q.x += dir.x; q.y += dir.y; q.z += dir.z;
    }
  }
  for (int i = 0; i < wave_steps; i++) {
    bool need = false;    // this lane asks for slot h at this step
    int h = 0;
    if (i < no_steps) {
      const short bx = (short)(int)floorf(pt.x), by = (short)(int)floorf(pt.y), bz = (short)(int)floorf(pt.z);
      h = hash_index(bx, by, bz, p.mask);
      HashEntry e = i == 0 ? pre[0] : (i == 1 ? pre[1] : (i == 2 ? pre[2] : load_entry(p.hash, h)));
      bool found = false;
      if (e.pos[0] == bx && e.pos[1] == by && e.pos[2] == bz && e.ptr >= -1) {
        p.vis_type[h] = (unsigned char)(p.gen | ((e.ptr == -1) ? 2u : 1u));
        found = true;
      }
      if (!found) {
        if (e.ptr >= -1) {
          while (e.offset >= 1) {
            h = p.num_buckets + e.offset - 1;
            e = load_entry(p.hash, h);
            if (e.pos[0] == bx && e.pos[1] == by && e.pos[2] == bz && e.ptr >= -1) {
              p.vis_type[h] = (unsigned char)(p.gen | ((e.ptr == -1) ? 2u : 1u));
              found = true;
              break;
            }
          }
        }
        need = !found;
      }
      pt.x += dir.x; pt.y += dir.y; pt.z += dir.z;
    }
    const unsigned key = (unsigned)idx * (unsigned)p.step_cap + (unsigned)i + 1u;
    // Neighbouring pixels ask for the same slot (a block covers hundreds of pixels), and same-address atomics
    // serialise at ~10 ns each on this part.  Keys grow with the pixel index, so within a wavefront the highest
    // lane of each group of equal slots holds the group's maximum: only that lane issues the atomicMax.
    unsigned long long todo = __ballot(need);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int hl = __shfl(h, leader, 64);
      const unsigned long long grp = __ballot(need && h == hl);
      if (lane == 63 - __clzll((long long)grp)) atomicMax(&p.keys[h], key);
      todo &= ~grp;
    }
  }
}

struct SweepParams {
  HashEntry *hash;
  int n_entries, num_buckets, n_tiles;
  unsigned *keys;
  unsigned char *alloc_type;
  short4 *coords;
  int *req_list;
  int *req_count;
  const int *alloc_list;
  const int *excess_list;
  unsigned char *vis_type;
  unsigned char *swap_state;
  SceneCounters *cnt;
  RenderCounters *rc;
  int *visible_ids;
  int capacity;
  unsigned long long *agg_req, *agg_succ, *agg_vis;
  unsigned epoch;
  unsigned gen;
  int do_commit;
  // replay of a winner's walk
  const float *depth;
  int W, H;
  Mat4 invM;
  float inv_fx, inv_fy, cx, cy, mu, one_over_block;
  int cap_shift;  // step_cap = 1 << cap_shift
  unsigned long long *dbg;  // diagnostics (DSLAM_DBG_SWEEP=<file>): per tile 8 timestamps
  // frustum re-test of the entries that were visible in the previous pass
  Mat4 M;
  float fx, fy, voxel_size;
};

// tile_block_vis for kSweepPer entries per thread: out[k] bit 0 = visible, bit 1 = visible in the enlarged frustum
struct SweepVisScratch {
  short4 pos[kSweepTile];
  unsigned short idx[kSweepTile];
  unsigned char res[kSweepTile];
  int n;
};

template <bool SWAPPING>
__device__ __forceinline__ void sweep_block_vis(SweepVisScratch &s, unsigned cand_mask, const HashEntry *__restrict__ hash,
                                                int t0, const Mat4 &M, float fx, float fy, float cx, float cy,
                                                float voxel_size, int W, int H, unsigned char out[kSweepPer]) {
  if (threadIdx.x == 0) s.n = 0;
#pragma unroll
  for (int q = 0; q < kSweepPer / 4; q++) *reinterpret_cast<unsigned *>(&s.res[threadIdx.x * kSweepPer + q * 4]) = 0u;
  __syncthreads();
  for (unsigned m = cand_mask; m; m &= m - 1) {
    const int k = __ffs((int)m) - 1;
    const HashEntry e = load_entry(hash, t0 + k);
    const int j = atomicAdd(&s.n, 1);
    s.pos[j] = make_short4(e.pos[0], e.pos[1], e.pos[2], 0);
    s.idx[j] = (unsigned short)(threadIdx.x * kSweepPer + k);
  }
  __syncthreads();
  const int n = s.n;
  for (int j = threadIdx.x; j < n; j += 256) {
    const short4 b = s.pos[j];
    bool vis, vis_enl;
    check_block_vis<SWAPPING>(vis, vis_enl, b.x, b.y, b.z, M, fx, fy, cx, cy, voxel_size, W, H);
    s.res[s.idx[j]] = (unsigned char)((vis ? 1 : 0) | (vis_enl ? 2 : 0));
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kSweepPer / 4; q++) {
    const unsigned r = *reinterpret_cast<const unsigned *>(&s.res[threadIdx.x * kSweepPer + q * 4]);
    out[q * 4] = r & 0xff; out[q * 4 + 1] = (r >> 8) & 0xff; out[q * 4 + 2] = (r >> 16) & 0xff; out[q * 4 + 3] = r >> 24;
  }
}

// Per tile: [A] winners -> requests (publish) -> ranks (look-back) -> commit (publish results);
//           [B] settle the entries that were visible before (frustum re-test), count (publish), ranks (look-back), list.
// B's loads and its frustum test do not depend on A's result, so a workgroup that owns ONE tile (the normal case: the grid
// covers the table) runs them while it waits for A's look-back and keeps the outcome in registers; its visible count goes
// out right after its own commits.  Only the tiles of the excess area, where OTHER tiles' commits create entries, have to
// wait for every tile's commit word before they can count.  A workgroup that owns several tiles (a table larger than
// the resident grid) does A for all of them, then B for all of them, so that no tile waits for one its own workgroup
// has not started.
template <bool SWAPPING>
__global__ __launch_bounds__(256) void k_alloc_sweep(SweepParams p) {
  __shared__ int red[2][8];
  __shared__ SweepVisScratch vis_scratch;
  // the pool tops are not modified before every tile has finished committing (the last tile folds the results in)
  const int base_free = p.cnt->last_free, base_free_ex = p.cnt->last_free_ex;
  const int avail_vba = base_free + 1, avail_ex = base_free_ex + 1;
  const bool single = (int)gridDim.x >= p.n_tiles;  // every workgroup owns exactly one tile
  int all_requests = 0;                             // (kept by the workgroup that owns the last tile)
  // state a single-tile workgroup carries from A to B
  unsigned char v[kSweepPer], ty[kSweepPer];
  int vis_rank = 0, vis_tot = 0;
  bool vis_published = false;
#define STAMP(i) if (p.dbg && threadIdx.x == 0) p.dbg[(size_t)b * 8 + (i)] = __builtin_amdgcn_s_memtime()

  // B, first half: types of this tile's entries after the pass (0: not visible) and their number.  `reload`: read the
  // bytes (again) -- everything that can write them from outside this workgroup has been waited for.
  auto load_vis = [&](int t0, bool in) {
#pragma unroll
    for (int k = 0; k < kSweepPer; k++) v[k] = 0;
    if (in) {
#pragma unroll
      for (int q = 0; q < kSweepPer / 4; q++) {
        const uchar4 v4 = *reinterpret_cast<const uchar4 *>(p.vis_type + t0 + q * 4);
        v[q * 4] = v4.x; v[q * 4 + 1] = v4.y; v[q * 4 + 2] = v4.z; v[q * 4 + 3] = v4.w;
      }
    }
  };
  auto settle = [&](int b, int t0, bool in) {
    unsigned cand_mask = 0;
#pragma unroll
    for (int k = 0; k < kSweepPer; k++)  // visible in the previous pass, not marked in this one: upstream's type 3
      if (v[k] != 0 && (v[k] & 0x80u) != p.gen) cand_mask |= 1u << k;
    unsigned char f[kSweepPer];
    sweep_block_vis<SWAPPING>(vis_scratch, cand_mask, p.hash, t0, p.M, p.fx, p.fy, p.cx, p.cy, p.voxel_size, p.W, p.H, f);
#pragma unroll
    for (int k = 0; k < kSweepPer; k++) {
      ty[k] = 0;
      if (v[k] == 0) continue;
      if ((cand_mask >> k) & 1u) ty[k] = (f[k] & (SWAPPING ? 2 : 1)) ? 3 : 0;
      else ty[k] = v[k] & 0x7f;
    }
  };
  // count the visible entries of the tile, publish the count
  auto count_and_publish = [&](int b, int t0) {
    int c = 0;
#pragma unroll
    for (int k = 0; k < kSweepPer; k++) {
      if (SWAPPING && ty[k] > 0 && p.swap_state[t0 + k] != 2) p.swap_state[t0 + k] = 1;
      c += ty[k] > 0;
    }
    vis_rank = block_excl_scan<4>(c, red[0], vis_tot);
    if (threadIdx.x == 0) publish(p.agg_vis, b, p.epoch, vis_tot >> 12, vis_tot & 0xfff);  // (<= 4096, as two fields)
  };

  // ---- A: winners -> requests -> commit -----------------------------------------------------------------------
  for (int b = blockIdx.x; b < p.n_tiles; b += gridDim.x) {
    STAMP(0);
    const int t0 = b * kSweepTile + threadIdx.x * kSweepPer;
    const bool in = t0 < p.n_entries;  // (entry counts are multiples of 16: a thread's entries are all inside or all outside)
    const bool has_excess = (b + 1) * kSweepTile > p.num_buckets;  // other tiles' commits may create entries in this one
    unsigned amask = 0;  // 2 bits per entry: 0 none, 1 ordered request, 2 excess request
    int c1 = 0, c2 = 0;
    if (in) {
      uint4 kk[kSweepPer / 4];
#pragma unroll
      for (int q = 0; q < kSweepPer / 4; q++) kk[q] = *reinterpret_cast<const uint4 *>(p.keys + t0 + q * 4);
      if (single) load_vis(t0, in);  // (in flight together with the keys)
#pragma unroll
      for (int q = 0; q < kSweepPer / 4; q++) {
        if (!(kk[q].x | kk[q].y | kk[q].z | kk[q].w)) continue;
        *reinterpret_cast<uint4 *>(p.keys + t0 + q * 4) = make_uint4(0, 0, 0, 0);  // leave the keys clean for the next pass
        const unsigned key[4] = {kk[q].x, kk[q].y, kk[q].z, kk[q].w};
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (key[k]) {
            // replay the winning pixel's walk up to the winning step: the block it asked for
            const int t = t0 + q * 4 + k;
            const unsigned kz = key[k] - 1u;
            const int pix = (int)(kz >> p.cap_shift), step = (int)(kz & ((1u << p.cap_shift) - 1u));
            const int py = pix / p.W, px = pix - py * p.W;
            Vec3 pt, dir;
            ray_segment(p.depth[pix], px, py, p.invM, p.inv_fx, p.inv_fy, p.cx, p.cy, p.mu, p.one_over_block, pt, dir);
            for (int i = 0; i < step; i++) { pt.x += dir.x; pt.y += dir.y; pt.z += dir.z; }
            // the key sits on an empty bucket head (ordered request) or on the last entry of an occupied bucket's
            // chain (excess request)
            const unsigned a = (p.hash[t].ptr >= -1) ? 2u : 1u;
            p.alloc_type[t] = (unsigned char)a;
            p.coords[t] = make_short4((short)(int)floorf(pt.x), (short)(int)floorf(pt.y), (short)(int)floorf(pt.z), 1);
            amask |= a << (2 * (q * 4 + k));
            c1 += a == 1u;
            c2 += a == 2u;
          }
      }
    }
    int tot1, tot2;
    int r1 = block_excl_scan<4>(c1, red[0], tot1);
    int r2 = block_excl_scan<4>(c2, red[1], tot2);
    if (threadIdx.x == 0) publish(p.agg_req, b, p.epoch, tot1, tot2);
    STAMP(1);
    // (single-tile workgroups) B's frustum test while the words in front of this tile arrive
    if (single) {
      if (!in) load_vis(t0, in);
      settle(b, t0, in);
    }
    const bool last = b == p.n_tiles - 1;
    int succ_vba = 0, succ_ex = 0;
    bool remote = false;  // stores into another tile's entries (a new excess entry)
    bool succ_published = false;
    if (tot1 + tot2 > 0 || last) {
      int pre1, pre2;
      lookback(p.agg_req, b, p.epoch, red[0], pre1, pre2);
      if (last) {
        all_requests = pre1 + tot1 + pre2 + tot2;
        if (threadIdx.x == 0) *p.req_count = all_requests;
      }
      int rr = pre1 + pre2 + r1 + r2;  // position in the list of this pass' requests (any unique position will do)
      r1 += pre1;
      r2 += pre2;
      // Which requests get a block follows from the ranks alone, so the tile's commit word can go out before its
      // stores -- unless it creates entries in another tile (an excess request that succeeds): that store has to be
      // visible device-wide before the word is, because the tiles of the excess area read it after seeing the word.
      unsigned okmask = 0;
      {
        int q1 = r1, q2 = r2;
        for (unsigned m = amask; m; ) {
          const int k = (__ffs((int)m) - 1) >> 1;
          const unsigned a = (amask >> (2 * k)) & 3u;
          m &= ~(3u << (2 * k));
          // voxel-block slots consumed by all earlier requests in hash-index order (closed form)
          const int vr = q1 + (q2 < avail_ex ? q2 : avail_ex);
          if (a == 1u) {
            if (p.do_commit && vr < avail_vba) { okmask |= 1u << k; succ_vba++; }
            q1++;
          } else {
            if (p.do_commit && q2 < avail_ex && vr < avail_vba) { okmask |= 1u << k; succ_vba++; succ_ex++; remote = true; }
            q2++;
          }
        }
      }
      for (int pass = 0; pass < 2; pass++) {  // pass 0: the stores into other tiles; pass 1: the rest
        if (pass == 1 && p.do_commit) {
          for (int d = 32; d > 0; d >>= 1) { succ_vba += __shfl_xor(succ_vba, d, 64); succ_ex += __shfl_xor(succ_ex, d, 64); }
          const unsigned long long any_remote = __ballot(remote);
          if (any_remote) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          if ((threadIdx.x & 63) == 0) {
            red[1][threadIdx.x >> 6] = succ_vba;
            red[1][4 + (threadIdx.x >> 6)] = succ_ex;
          }
          __syncthreads();
          if (threadIdx.x == 0) {
            // (a tile that stored into other tiles: each of its storing waves waited for its write-through stores in
            // front of the barrier above; no release fence -- see pass 0)
            publish(p.agg_succ, b, p.epoch, red[1][0] + red[1][1] + red[1][2] + red[1][3], red[1][4] + red[1][5] + red[1][6] + red[1][7]);
          }
          succ_published = true;
        }
        int q1 = r1, q2 = r2, qq = rr;
        for (unsigned m = amask; m; ) {
          const int k = (__ffs((int)m) - 1) >> 1;
          const unsigned a = (amask >> (2 * k)) & 3u;
          m &= ~(3u << (2 * k));
          const int t = t0 + k;
          const int vr = q1 + (q2 < avail_ex ? q2 : avail_ex);
          const bool ok = (okmask >> k) & 1u;
          if (a == 1u) {
            if (pass == 1) {
              p.req_list[qq] = t;
              if (ok) {
                const short4 bc = p.coords[t];
                store_entry(p.hash, t, bc.x, bc.y, bc.z, 0, p.alloc_list[base_free - vr]);
              }
              // without the commit (onlyUpdateVisibleList) the request alone makes the entry "visible" this pass, like
              // upstream; with it, only if it got a block
              const unsigned char nv = (ok || !p.do_commit) ? (unsigned char)(p.gen | 1u) : (unsigned char)0;
              p.vis_type[t] = nv;
              if (single) { v[k] = nv; ty[k] = nv & 0x7f; }  // (whatever an empty bucket head carried before)
            }
            q1++;
          } else {
            if (pass == 0 && ok) {
              const int ex_off = p.excess_list[base_free_ex - q2];
              const short4 bc = p.coords[t];
              p.hash[t].offset = ex_off + 1;
              // The two stores that land in ANOTHER tile's entries go out write-through (sc1): once this wave's vmcnt is
              // back at 0 they are in memory, so the commit word needs no agent-scope release in front of it -- that is a
              // write-back of the XCD's L2 (1.7-6.5 us) on the critical path of every tile behind this one (the tiles of
              // the excess area wait for EVERY commit word): last tile done at 17.5 instead of 19.3 us.  The readers keep
              // their agent-scope acquire (MI355X_MICROARCH.md, inter-workgroup visibility: sc1 stores + the storing
              // waves' vmcnt(0) + barrier, then the flag; acquire + plain loads on the other side).
              {  // (relaxed agent-scope atomic stores = plain store instructions with sc1; the entry as two 8-byte halves)
                unsigned long long *dst = reinterpret_cast<unsigned long long *>(p.hash + (p.num_buckets + ex_off));
                const unsigned long long lo = ((unsigned long long)((unsigned)bc.z & 0xffffu) << 32) |
                                              (((unsigned)bc.x & 0xffffu) | ((unsigned)bc.y << 16));
                const unsigned long long hi = (unsigned long long)(unsigned)p.alloc_list[base_free - vr] << 32;  // offset 0, ptr
                __hip_atomic_store(dst, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(dst + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(p.vis_type + (p.num_buckets + ex_off), (unsigned char)(p.gen | 1u), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
              }
            }
            if (pass == 1) p.req_list[qq] = t;
            q2++;
          }
          qq++;
        }
      }
    }
    // a tile without excess-area entries knows its visible entries now
    vis_published = false;
    if (single && !(p.do_commit && has_excess)) {
      count_and_publish(b, t0);
      vis_published = true;
    }
    if (p.do_commit && !succ_published) {  // a tile without requests: nothing committed
      if (threadIdx.x == 0) publish(p.agg_succ, b, p.epoch, 0, 0);
    }
    STAMP(2);
  }

  // (a workgroup that owns several tiles may have created an entry in one of its own later tiles)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- B: settle visibility, build the visible list -------------------------------------------------------------
  for (int b = blockIdx.x; b < p.n_tiles; b += gridDim.x) {
    const int t0 = b * kSweepTile + threadIdx.x * kSweepPer;
    const bool in = t0 < p.n_entries;
    const bool last = b == p.n_tiles - 1;
    const bool has_excess = (b + 1) * kSweepTile > p.num_buckets;
    STAMP(3);
    int succ_vba_all = 0, succ_ex_all = 0;
    if (p.do_commit && (has_excess || last)) {
      lookback(p.agg_succ, p.n_tiles, p.epoch, red[0], succ_vba_all, succ_ex_all);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    STAMP(4);
    if (!vis_published) {
      if (!single) {
        load_vis(t0, in);
        settle(b, t0, in);
      } else if (p.do_commit && has_excess && in) {
        // other tiles' commits may have created entries here: they carry this pass' mark; everything else was settled in A.
        // A byte that differs from what this tile last saw or wrote IS such a mark -- also where the slot still carried
        // a stale type of an entry that no longer exists (a render state that outlived a ResetScene: the stale byte
        // used to hide the new entry's mark; fuzz seed 70473)
#pragma unroll
        for (int q = 0; q < kSweepPer / 4; q++) {
          const uchar4 v4 = *reinterpret_cast<const uchar4 *>(p.vis_type + t0 + q * 4);
          const unsigned char nb4[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
          for (int kk = 0; kk < 4; kk++)
            if (nb4[kk] != v[q * 4 + kk]) { v[q * 4 + kk] = nb4[kk]; ty[q * 4 + kk] = nb4[kk] & 0x7f; }
        }
      }
      count_and_publish(b, t0);
    }
    STAMP(5);
    int offset = 0;
    if (vis_tot > 0 || last) {
      int hi, lo;
      lookback(p.agg_vis, b, p.epoch, red[0], hi, lo);
      offset = hi * 4096 + lo;
    }
    STAMP(6);
    if (in) {
      int r = vis_rank + offset;
#pragma unroll
      for (int q = 0; q < kSweepPer / 4; q++) {
        bool changed = false;
        unsigned char nv4[4];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          const int k = q * 4 + kk;
          unsigned char nv = 0;
          if (ty[k] > 0) {
            if (r < p.capacity) {
              p.visible_ids[r] = t0 + k;
              nv = (unsigned char)(p.gen | ty[k]);
            } else {
              // no room in the list: upstream leaves the type in place without the entry being re-armed next pass, so
              // a 1 / 2 counts as marked again then (next pass' bit), a 3 is re-tested (this pass' bit)
              nv = (unsigned char)((ty[k] == 3 ? p.gen : (p.gen ^ 0x80u)) | ty[k]);
            }
            r++;
          }
          changed |= nv != v[k];
          nv4[kk] = nv;
        }
        if (changed) *reinterpret_cast<uchar4 *>(p.vis_type + t0 + q * 4) = make_uchar4(nv4[0], nv4[1], nv4[2], nv4[3]);
      }
    }
    if (last && threadIdx.x == 0) {
      const int n = offset + vis_tot;
      p.rc->no_visible = n < p.capacity ? n : p.capacity;
      if (p.do_commit) {
        p.cnt->last_free = base_free - succ_vba_all;
        p.cnt->last_free_ex = base_free_ex - succ_ex_all;
        p.cnt->alloc_failures = all_requests - succ_vba_all;
      } else {
        p.cnt->alloc_failures = 0;
      }
    }
    vis_published = false;
    STAMP(7);
  }
#undef STAMP
}

// The visible list of a render state was replaced behind its types' back (FindVisibleBlocks into this render state, an
// uploaded list).  Upstream's next pass would leave every type as it is and set the LIST's entries to 3; in the
// generation encoding: a 1 / 2 that is to stay "marked" gets the coming pass' bit, a 3 keeps the old bit (re-tested),
// and the list's entries become old-bit 3s.
__global__ __launch_bounds__(256) void k_types_keep(unsigned char *vis_type, int n_entries, unsigned new_gen) {
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n_entries) return;
  uchar4 v = *reinterpret_cast<uchar4 *>(vis_type + i);
  auto fix = [&](unsigned char x) -> unsigned char {
    const unsigned t = x & 0x7fu;
    if (t == 0) return 0;
    return (unsigned char)((t == 3 ? (new_gen ^ 0x80u) : new_gen) | t);
  };
  v.x = fix(v.x); v.y = fix(v.y); v.z = fix(v.z); v.w = fix(v.w);
  *reinterpret_cast<uchar4 *>(vis_type + i) = v;
}
__global__ __launch_bounds__(256) void k_types_rearm(const int *__restrict__ ids, const RenderCounters *rc,
                                                     unsigned char *vis_type, unsigned old_gen) {
  const int n = rc->no_visible;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) vis_type[ids[i]] = (unsigned char)(old_gen | 3u);
}

// reallocate swapped-out blocks that came back into view (useSwapping only): one pool, so the r-th request in
// hash-index order succeeds iff r < available
__global__ __launch_bounds__(256) void k_realloc_count(const unsigned char *__restrict__ vis_type,
                                                       const HashEntry *__restrict__ hash, int n_entries,
                                                       int *__restrict__ tile_counts, SceneCounters *cnt) {
  __shared__ int red[4];
  const int t0 = blockIdx.x * kTileEntries + threadIdx.x * 4;
  if (blockIdx.x == 0 && threadIdx.x == 0) cnt->base_free = cnt->last_free;  // (nothing moves the top before the apply pass ends)
  int c = 0;
  if (t0 < n_entries) {
    const uchar4 v = *reinterpret_cast<const uchar4 *>(vis_type + t0);
    const unsigned char f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (f[k] > 0 && hash[t0 + k].ptr == -1) c++;
  }
  int tot;
  block_excl_scan<4>(c, red, tot);
  if (threadIdx.x == 0) tile_counts[blockIdx.x] = tot;
}

// the tile's exclusive offset is the sum of the preceding tile counts, computed here; the last tile takes the slots
// off the pool top (every tile reads the top as it was from base_free)
__global__ __launch_bounds__(256) void k_realloc_apply(const unsigned char *__restrict__ vis_type, HashEntry *hash,
                                                       int n_entries, const int *__restrict__ tile_counts,
                                                       const int *__restrict__ alloc_list, SceneCounters *cnt) {
  __shared__ int red[4];
  const int t0 = blockIdx.x * kTileEntries + threadIdx.x * 4;
  bool need[4] = {false, false, false, false};
  int c = 0;
  if (t0 < n_entries) {
    const uchar4 v = *reinterpret_cast<const uchar4 *>(vis_type + t0);
    const unsigned char f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      need[k] = f[k] > 0 && hash[t0 + k].ptr == -1;
      c += need[k];
    }
  }
  int tot;
  int r = block_excl_scan<4>(c, red, tot);
  const bool last = blockIdx.x == gridDim.x - 1;
  if (tot == 0 && !last) return;
  const int offset = block_sum_strided(tile_counts, blockIdx.x, 1, red);
  const int base_free = cnt->base_free;
  if (last && threadIdx.x == 0) {
    const int total = offset + tot, avail = base_free + 1;
    cnt->last_free = base_free - (total < avail ? total : avail);
  }
  r += offset;
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (need[k]) {
      if (r <= base_free) hash[t0 + k].ptr = alloc_list[base_free - r];
      r++;
    }
}

static inline int ceil_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// workgroups of a sweep kernel that are certainly resident together (its tiles wait for each other inside the launch)
template <typename K>
static int resident_grid(dslam_engine *e, K kernel) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
  if (per_cu > 8) per_cu = 8;
  if (per_cu > 1) per_cu -= 1;  // (the occupancy query can be one block per CU high on this part: keep a margin)
  return per_cu * (e->sm_count > 0 ? e->sm_count : 1);
}

int launch_allocate(dslam_engine *e, dslam_scene *s, const dslam_view *v, dslam_render_state *r, const float *M_d,
                    const float *intr, int only_update_visible_list) {
  const int W = v->w_d, H = v->h_d, N = s->n_entries;
  DSLAM_REQUIRE(r->n_entries == N, "render state was created for a different scene size");
  DSLAM_REQUIRE((N & 15) == 0, "num_buckets + num_excess must be a multiple of 16");
  int rc = ensure_scratch(e, N, s->p.num_local_blocks);
  if (rc) return rc;
  if (e->sweep_grid_cap == 0) {
    const int a = resident_grid(e, k_alloc_sweep<false>), b = resident_grid(e, k_alloc_sweep<true>);
    e->sweep_grid_cap = a < b ? a : b;
  }

  MarkParams mp;
  mp.raw = v->depth_dirty ? v->raw_src : nullptr;
  mp.depth = v->depth; mp.a = v->affine_a; mp.b = v->affine_b;
  mp.W = W; mp.H = H;
  if (!invert_matrix(M_d, mp.invM.m)) { set_last_error("pose matrix is singular"); return DSLAM_ERR_INVALID; }
  mp.inv_fx = 1.0f / intr[0]; mp.inv_fy = 1.0f / intr[1]; mp.cx = intr[2]; mp.cy = intr[3];
  mp.mu = s->p.mu; mp.frustum_min = s->p.frustum_min; mp.frustum_max = s->p.frustum_max;
  mp.one_over_block = 1.0f / (s->p.voxel_size * kBlock);
  mp.hash = s->hash; mp.mask = (unsigned)(s->p.num_buckets - 1); mp.num_buckets = s->p.num_buckets;
  mp.keys = e->order_keys; mp.vis_type = r->visible_type;
  mp.cnt = s->counters;
  mp.alloc_type = e->alloc_type; mp.req_list = e->req_list; mp.req_count = e->req_count;
  // steps along the +-mu segment: ceil(2 * |segment| in blocks) = ceil(mu / (2 * voxel_size)) for a rigid pose
  const int step_bound = (int)ceilf(s->p.mu / (2.0f * s->p.voxel_size)) + 2;
  mp.step_cap = ceil_pow2(step_bound + 1);
  if ((double)W * H * mp.step_cap >= 4294967295.0) {
    set_last_error("image size x ray steps exceeds the 32-bit allocation order key");
    return DSLAM_ERR_UNSUPPORTED;
  }
  int cap_shift = 0;
  while ((1 << cap_shift) < mp.step_cap) cap_shift++;

  // this pass' generation bit; what the previous pass left visible carries the other one
  const unsigned old_gen = r->gen;
  r->gen ^= 0x80u;
  mp.gen = r->gen;
  if (!r->types_follow_list) {
    hipLaunchKernelGGL(k_types_keep, dim3((N / 4 + 255) / 256), dim3(256), 0, e->stream, r->visible_type, N, (unsigned)r->gen);
    hipLaunchKernelGGL(k_types_rearm, dim3(64), dim3(256), 0, e->stream, r->visible_ids, r->counters, r->visible_type, old_gen);
    r->types_follow_list = true;
  }

  const int pix_blocks = (W * H + 255) / 256;
  hipLaunchKernelGGL(k_mark, dim3(pix_blocks), dim3(256), 0, e->stream, mp);
  v->depth_dirty = false;

  const int n_tiles = (N + kSweepTile - 1) / kSweepTile;
  SweepParams sp;
  sp.hash = s->hash; sp.n_entries = N; sp.num_buckets = s->p.num_buckets; sp.n_tiles = n_tiles;
  sp.keys = e->order_keys; sp.alloc_type = e->alloc_type; sp.coords = e->block_coords;
  sp.req_list = e->req_list; sp.req_count = e->req_count;
  sp.alloc_list = s->alloc_list; sp.excess_list = s->excess_list;
  sp.vis_type = r->visible_type; sp.swap_state = s->swap_state;
  sp.cnt = s->counters; sp.rc = r->counters; sp.visible_ids = r->visible_ids; sp.capacity = r->n_local;
  sp.agg_req = e->agg; sp.agg_succ = e->agg + e->agg_tiles; sp.agg_vis = e->agg + 2 * (size_t)e->agg_tiles;  // (sized for 1024-entry tiles)
  if (++e->epoch == 0) e->epoch = 1;
  sp.epoch = e->epoch;
  sp.gen = r->gen;
  sp.do_commit = only_update_visible_list ? 0 : 1;
  sp.depth = v->depth; sp.W = W; sp.H = H;
  sp.invM = mp.invM; sp.inv_fx = mp.inv_fx; sp.inv_fy = mp.inv_fy; sp.cx = mp.cx; sp.cy = mp.cy;
  sp.mu = mp.mu; sp.one_over_block = mp.one_over_block; sp.cap_shift = cap_shift;
  memcpy(sp.M.m, M_d, sizeof(float) * 16);
  sp.fx = intr[0]; sp.fy = intr[1]; sp.voxel_size = s->p.voxel_size;
  const int grid = n_tiles < e->sweep_grid_cap ? n_tiles : e->sweep_grid_cap;
  sp.dbg = nullptr;
  static const char *dbg_file = getenv("DSLAM_DBG_SWEEP");
  static int dbg_calls = 0;
  unsigned long long *dbg_host = nullptr;
  if (dbg_file && ++dbg_calls == 60) {
    DSLAM_HIP(hipHostMalloc((void **)&dbg_host, (size_t)n_tiles * 64, hipHostMallocDefault));
    memset(dbg_host, 0, (size_t)n_tiles * 64);
    sp.dbg = dbg_host;
  }
  if (s->p.use_swapping) hipLaunchKernelGGL(k_alloc_sweep<true>, dim3(grid), dim3(256), 0, e->stream, sp);
  else hipLaunchKernelGGL(k_alloc_sweep<false>, dim3(grid), dim3(256), 0, e->stream, sp);
  if (s->p.use_swapping) {
    const int r_tiles = num_tiles(N);
    hipLaunchKernelGGL(k_realloc_count, dim3(r_tiles), dim3(256), 0, e->stream, r->visible_type, s->hash, N,
                       e->tile_counts, s->counters);
    hipLaunchKernelGGL(k_realloc_apply, dim3(r_tiles), dim3(256), 0, e->stream, r->visible_type, s->hash, N,
                       e->tile_counts, s->alloc_list, s->counters);
  }
  DSLAM_HIP(hipGetLastError());
  if (dbg_host) {
    DSLAM_HIP(hipStreamSynchronize(e->stream));
    FILE *f = fopen(dbg_file, "wb");
    if (f) { fwrite(dbg_host, 64, n_tiles, f); fclose(f); }
    (void)hipHostFree(dbg_host);
  }
  return DSLAM_OK;
}

// ---------------------------------------------------------------------------------------------------------
// visible-list history: set this list's ring bit on every resident visible block
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_clear_ring_bit(unsigned long long *masks, int n_local, int words, int ring,
                                                        int bit) {
  const unsigned long long m = ~(1ull << (bit & 63));
  for (int slot = blockIdx.x * blockDim.x + threadIdx.x; slot < n_local; slot += gridDim.x * blockDim.x)
    masks[((size_t)slot * 2 + ring) * words + (bit >> 6)] &= m;
}

// host side of queueing a visible list on ring q: ring bookkeeping (+ the rare drop of the oldest list of a full
// ring); the bit itself is set by the integrate kernel for every visible block it visits
int prepare_push_visible_list(dslam_engine *e, dslam_scene *s, int q, int *bit_out, int *frame_out) {
  const int bits = 64 * s->history_words;
  if (s->ring_next[q] - s->ring_head[q] == bits) {  // full ring: drop the oldest list, release nothing
    hipLaunchKernelGGL(k_clear_ring_bit, dim3(512), dim3(256), 0, e->stream, s->masks, s->p.num_local_blocks,
                       s->history_words, q, s->ring_head[q] % bits);
    DSLAM_HIP(hipGetLastError());
    s->ring_head[q]++;
    if (s->decay_cursor[q] < s->ring_head[q]) s->decay_cursor[q] = s->ring_head[q];
  }
  *bit_out = (s->ring_next[q]++) % bits;
  *frame_out = s->frame_counter++;
  return DSLAM_OK;
}

}  // namespace dslam
