// alloc.hip -- scene reset, view conversion and AllocateSceneFromDepth for gfx950.
//
// Reference call sites: denseMapper->ResetScene (InfiniTamDriver.h:354-360), viewBuilder->UpdateView
// (InfiniTamDriver.cpp:280-288), denseMapper->ProcessFrame -> AllocateSceneFromDepth (InfiniTamDriver.h:187-192).
// Algorithm: SURVEY.md Appendix A.3, A.4, A.10.
//
// Allocation is bit-exact with the sequential CPU engine ("last writer in row-major pixel order wins", pool
// slots handed out in ascending hash-index order) although it runs wave-parallel:
//   mark A   per pixel, walk the +-mu segment in block units; misses do atomicMax(order_key[slot], pixel*cap+step+1)
//   mark B   the same walk; only the (pixel, step) that owns the key writes allocType / blockCoords
//   commit   count / scan / apply over tiles of 1024 entries: the r-th requesting entry in hash-index order gets
//            voxelAllocationList[lastFree - r]; pool exhaustion follows the closed form derived in DESIGN.md
//   visible  count (type-3 frustum re-test) / scan / apply: visibleEntryIDs ascending in hash index
// No host round trip between the phases: every count lives in device memory (SceneCounters/RenderCounters).
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
// One preparation kernel for the allocation pass (four independent jobs, one launch): clear the order keys +
// allocType scratch, zero the commit tile counters, re-arm the previous visible list as type 3, and -- if the view
// was updated since -- derive the float depth image from the raw millimetre image.
__global__ __launch_bounds__(256) void k_alloc_prep(uint4 *scratch16, int scratch_n16, int *tile_counts, int n_counts,
                                                    const int *__restrict__ visible_ids, const RenderCounters *rc,
                                                    unsigned char *vis_type, const short *__restrict__ raw,
                                                    float *__restrict__ depth, int npix, float a, float b) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  const uint4 z = make_uint4(0, 0, 0, 0);
  for (int i = tid; i < scratch_n16; i += stride) scratch16[i] = z;
  for (int i = tid; i < n_counts; i += stride) tile_counts[i] = 0;
  const int n = rc->no_visible;
  for (int i = tid; i < n; i += stride) vis_type[visible_ids[i]] = 3;
  if (raw)
    for (int i = tid; i < npix; i += stride) {
      const int d = raw[i];
      depth[i] = (d <= 0 || d > 32000) ? -1.0f : (float)d * a + b;
    }
}

struct MarkParams {
  const float *depth;
  int W, H;
  Mat4 invM;
  float inv_fx, inv_fy, cx, cy;
  float mu, frustum_min, frustum_max, one_over_block;
  const HashEntry *hash;
  unsigned mask;
  int num_buckets;
  unsigned *keys;
  unsigned char *alloc_type;
  short4 *coords;
  unsigned char *vis_type;
  int step_cap;
  SceneCounters *cnt;
  int *tile_counts;  // per-tile (type 1, type 2) request counts, accumulated by the phase-1 winners
};

// buildHashAllocAndVisibleTypePP.  PHASE 0: found entries mark visibility, misses race for the slot's order
// key.  PHASE 1: the winner of each slot writes the allocation request.
template <int PHASE>
__global__ __launch_bounds__(256) void k_mark(MarkParams p) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  // every lane stays in the kernel (invalid pixels march zero steps): the order-key atomics below are aggregated
  // per wavefront, which needs the wave converged
  const bool in_image = idx < p.W * p.H;
  const int y = in_image ? idx / p.W : 0, x = in_image ? idx - y * p.W : 0;
  const float d = in_image ? p.depth[idx] : -1.0f;
  const bool valid = !(d <= 0 || (d - p.mu) < 0 || (d - p.mu) < p.frustum_min || (d + p.mu) > p.frustum_max);

  Vec3 pc;
  pc.z = d;
  pc.x = pc.z * (((float)x - p.cx) * p.inv_fx);
  pc.y = pc.z * (((float)y - p.cy) * p.inv_fy);
  float norm = sqrtf(pc.x * pc.x + pc.y * pc.y + pc.z * pc.z);
  Vec4 tmp;
  tmp.x = pc.x * (1.0f - p.mu / norm); tmp.y = pc.y * (1.0f - p.mu / norm); tmp.z = pc.z * (1.0f - p.mu / norm); tmp.w = 1.0f;
  Vec4 q = mul(p.invM, tmp);
  Vec3 pt = {q.x * p.one_over_block, q.y * p.one_over_block, q.z * p.one_over_block};
  tmp.x = pc.x * (1.0f + p.mu / norm); tmp.y = pc.y * (1.0f + p.mu / norm); tmp.z = pc.z * (1.0f + p.mu / norm);
  q = mul(p.invM, tmp);
  Vec3 pe = {q.x * p.one_over_block, q.y * p.one_over_block, q.z * p.one_over_block};
  Vec3 dir = {pe.x - pt.x, pe.y - pt.y, pe.z - pt.z};
  norm = sqrtf(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
  int no_steps = valid ? (int)ceilf(2.0f * norm) : 0;
  const float div = (float)(no_steps - 1);
  dir.x /= div; dir.y /= div; dir.z /= div;
  if (no_steps > p.step_cap) {  // the order key cannot encode later steps: report instead of mis-ordering
    if (PHASE == 0) atomicOr(&p.cnt->error_flags, 1);
    no_steps = p.step_cap;
  }
  int wave_steps = no_steps;
  for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(wave_steps, o, 64); wave_steps = v > wave_steps ? v : wave_steps; }

  for (int i = 0; i < wave_steps; i++) {
    bool need = false;    // this lane asks for slot h at this step
    bool excess = false;
    int h = 0;
    short bx = 0, by = 0, bz = 0;
    if (i < no_steps) {
      bx = (short)(int)floorf(pt.x); by = (short)(int)floorf(pt.y); bz = (short)(int)floorf(pt.z);
      h = hash_index(bx, by, bz, p.mask);
      HashEntry e = load_entry(p.hash, h);
      bool found = false;
      if (e.pos[0] == bx && e.pos[1] == by && e.pos[2] == bz && e.ptr >= -1) {
        if (PHASE == 0) p.vis_type[h] = (e.ptr == -1) ? 2 : 1;
        found = true;
      }
      if (!found) {
        if (e.ptr >= -1) {
          while (e.offset >= 1) {
            h = p.num_buckets + e.offset - 1;
            e = load_entry(p.hash, h);
            if (e.pos[0] == bx && e.pos[1] == by && e.pos[2] == bz && e.ptr >= -1) {
              if (PHASE == 0) p.vis_type[h] = (e.ptr == -1) ? 2 : 1;
              found = true;
              break;
            }
          }
          excess = true;
        }
        need = !found;
      }
      pt.x += dir.x; pt.y += dir.y; pt.z += dir.z;
    }
    const unsigned key = (unsigned)idx * (unsigned)p.step_cap + (unsigned)i + 1u;
    if (PHASE == 0) {
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
    } else if (need && p.keys[h] == key) {
      atomicAdd(&p.tile_counts[(h / kTileEntries) * 2 + (excess ? 1 : 0)], 1);  // commit pass 1, for free
      p.alloc_type[h] = excess ? 2 : 1;
      if (!excess) p.vis_type[h] = 1;
      p.coords[h] = make_short4(bx, by, bz, 1);
    }
  }
}

// ---- ordered compaction building blocks -------------------------------------------------------------------
// A tile = 1024 consecutive entries handled by one 256-thread workgroup, 4 consecutive entries per thread.

__global__ __launch_bounds__(256) void k_commit_apply(const unsigned char *__restrict__ alloc_type,
                                                      const short4 *__restrict__ coords, int n_entries,
                                                      const int *__restrict__ tile_counts, HashEntry *hash,
                                                      int num_buckets, const int *__restrict__ alloc_list,
                                                      const int *__restrict__ excess_list, unsigned char *vis_type,
                                                      SceneCounters *cnt, int n_tiles) {
  __shared__ int red[2][8];
  const int t0 = blockIdx.x * kTileEntries + threadIdx.x * 4;
  unsigned char a[4] = {0, 0, 0, 0};
  if (t0 < n_entries) {
    const uchar4 v = *reinterpret_cast<const uchar4 *>(alloc_type + t0);
    a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
  }
  int c1 = 0, c2 = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) { c1 += (a[k] == 1); c2 += (a[k] == 2); }
  int tot1, tot2;
  int r1 = block_excl_scan<4>(c1, red[0], tot1);
  int r2 = block_excl_scan<4>(c2, red[1], tot2);
  if (tot1 + tot2 == 0) return;
  // exclusive tile offsets = sums of the preceding tiles' request counts (accumulated by the mark pass)
  int pre1, pre2, all1, all2;
  block_prefix_and_total(tile_counts, blockIdx.x, n_tiles, 2, red[0], pre1, all1);
  block_prefix_and_total(tile_counts + 1, blockIdx.x, n_tiles, 2, red[1], pre2, all2);
  r1 += pre1;
  r2 += pre2;
  // the pool tops are not modified during this kernel (the next kernel folds the success counts into them)
  const int base_free = cnt->last_free, base_free_ex = cnt->last_free_ex;
  const int avail_vba = base_free + 1, avail_ex = base_free_ex + 1;
  // unless the voxel-block pool runs out during this pass, the success counts follow from the totals in closed
  // form (k_visible_count folds them in); only the exhausted regime counts them with (same-address) atomics
  const bool exhausted = all1 + (all2 < avail_ex ? all2 : avail_ex) > avail_vba;
  int succ_vba = 0, succ_ex = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int t = t0 + k;
    if (a[k] == 1 || a[k] == 2) {
      // voxel-block slots consumed by all earlier requests in hash-index order (closed form)
      const int vr = r1 + (r2 < avail_ex ? r2 : avail_ex);
      const short4 b = coords[t];
      if (a[k] == 1) {
        if (vr < avail_vba) {
          store_entry(hash, t, b.x, b.y, b.z, 0, alloc_list[base_free - vr]);
          succ_vba++;
        } else {
          vis_type[t] = 0;
        }
        r1++;
      } else {
        if (r2 < avail_ex && vr < avail_vba) {
          const int ex_off = excess_list[base_free_ex - r2];
          hash[t].offset = ex_off + 1;
          store_entry(hash, num_buckets + ex_off, b.x, b.y, b.z, 0, alloc_list[base_free - vr]);
          vis_type[num_buckets + ex_off] = 1;
          succ_vba++;
          succ_ex++;
        }
        r2++;
      }
    }
  }
  if (exhausted) {
    if (succ_vba) atomicAdd(&cnt->commit_succ_vba, succ_vba);
    if (succ_ex) atomicAdd(&cnt->commit_succ_ex, succ_ex);
  }
}

struct VisParams {
  Mat4 M;
  float fx, fy, cx, cy, voxel_size;
  int W, H;
};

// buildVisibleList, pass 1: settle type-3 entries (frustum re-test), swap-state bookkeeping, per-tile counts
template <bool SWAPPING>
__global__ __launch_bounds__(256) void k_visible_count(unsigned char *vis_type, const HashEntry *__restrict__ hash,
                                                       unsigned char *swap_state, int n_entries, VisParams p,
                                                       int *__restrict__ tile_counts, SceneCounters *cnt,
                                                       const int *__restrict__ commit_counts, int n_commit_counts,
                                                       int finalize_commit) {
  __shared__ int red[4];
  if (blockIdx.x == 0) {  // fold the commit pass' results into the pool tops (nobody else reads them in this kernel)
    const int all1 = finalize_commit ? block_sum_strided(commit_counts, n_commit_counts / 2, 2, red) : 0;
    const int all2 = finalize_commit ? block_sum_strided(commit_counts + 1, n_commit_counts / 2, 2, red) : 0;
    if (threadIdx.x == 0) {
      if (finalize_commit) {
        const int avail_vba = cnt->last_free + 1, avail_ex = cnt->last_free_ex + 1;
        const int ex_ok = all2 < avail_ex ? all2 : avail_ex;
        if (!(all1 + ex_ok > avail_vba)) {  // same test as k_commit_apply: nothing ran out of voxel blocks
          cnt->commit_succ_vba = all1 + ex_ok;
          cnt->commit_succ_ex = ex_ok;
        }
        const int requests = all1 + all2;
        cnt->last_free -= cnt->commit_succ_vba;
        cnt->last_free_ex -= cnt->commit_succ_ex;
        cnt->alloc_failures = requests - cnt->commit_succ_vba;
        cnt->commit_succ_vba = 0;
        cnt->commit_succ_ex = 0;
      } else {
        cnt->alloc_failures = 0;
      }
    }
  }
  __shared__ TileVisScratch vis_scratch;
  const int t0 = blockIdx.x * kTileEntries + threadIdx.x * 4;
  unsigned char v[4] = {0, 0, 0, 0};
  bool cand[4] = {false, false, false, false};
  short4 pos[4];
  if (t0 < n_entries) {
    const uchar4 v4 = *reinterpret_cast<const uchar4 *>(vis_type + t0);
    v[0] = v4.x; v[1] = v4.y; v[2] = v4.z; v[3] = v4.w;
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (v[k] == 3) {
        const HashEntry e = load_entry(hash, t0 + k);
        cand[k] = true;
        pos[k] = make_short4(e.pos[0], e.pos[1], e.pos[2], 0);
      }
  }
  unsigned char f[4];
  tile_block_vis<SWAPPING>(vis_scratch, cand, pos, p.M, p.fx, p.fy, p.cx, p.cy, p.voxel_size, p.W, p.H, f);
  int c = 0;
  if (t0 < n_entries) {
    bool changed = false;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (v[k] == 3 && !(f[k] & (SWAPPING ? 2 : 1))) { v[k] = 0; changed = true; }
      if (SWAPPING && v[k] > 0 && swap_state[t0 + k] != 2) swap_state[t0 + k] = 1;
      c += (v[k] > 0);
    }
    if (changed) *reinterpret_cast<uchar4 *>(vis_type + t0) = make_uchar4(v[0], v[1], v[2], v[3]);
  }
  int tot;
  block_excl_scan<4>(c, red, tot);
  if (threadIdx.x == 0) tile_counts[blockIdx.x] = tot;
}

// reallocate swapped-out blocks that came back into view (useSwapping only): one pool, so the r-th request in
// hash-index order succeeds iff r < available
__global__ __launch_bounds__(256) void k_realloc_count(const unsigned char *__restrict__ vis_type,
                                                       const HashEntry *__restrict__ hash, int n_entries,
                                                       int *__restrict__ tile_counts) {
  __shared__ int red[4];
  const int t0 = blockIdx.x * kTileEntries + threadIdx.x * 4;
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

__global__ __launch_bounds__(1024) void k_realloc_scan(const int *tile_counts, int *tile_offsets, int n_tiles,
                                                       SceneCounters *cnt) {
  int totals[1];
  scan_tiles<1>(tile_counts, tile_offsets, n_tiles, totals);
  if (threadIdx.x == 0) {
    cnt->base_free = cnt->last_free;
    const int avail = cnt->last_free + 1;
    const int got = totals[0] < avail ? totals[0] : avail;
    cnt->last_free -= got;
  }
}

__global__ __launch_bounds__(256) void k_realloc_apply(const unsigned char *__restrict__ vis_type, HashEntry *hash,
                                                       int n_entries, const int *__restrict__ tile_offsets,
                                                       const int *__restrict__ alloc_list, const SceneCounters *cnt) {
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
  if (tot == 0) return;
  r += tile_offsets[blockIdx.x];
  const int base_free = cnt->base_free;
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

int launch_allocate(dslam_engine *e, dslam_scene *s, const dslam_view *v, dslam_render_state *r, const float *M_d,
                    const float *intr, int only_update_visible_list) {
  const int W = v->w_d, H = v->h_d, N = s->n_entries;
  DSLAM_REQUIRE(r->n_entries == N, "render state was created for a different scene size");
  DSLAM_REQUIRE((N & 15) == 0, "num_buckets + num_excess must be a multiple of 16");
  int rc = ensure_scratch(e, N, s->p.num_local_blocks);
  if (rc) return rc;

  MarkParams mp;
  mp.depth = v->depth; mp.W = W; mp.H = H;
  if (!invert_matrix(M_d, mp.invM.m)) { set_last_error("pose matrix is singular"); return DSLAM_ERR_INVALID; }
  mp.inv_fx = 1.0f / intr[0]; mp.inv_fy = 1.0f / intr[1]; mp.cx = intr[2]; mp.cy = intr[3];
  mp.mu = s->p.mu; mp.frustum_min = s->p.frustum_min; mp.frustum_max = s->p.frustum_max;
  mp.one_over_block = 1.0f / (s->p.voxel_size * kBlock);
  mp.hash = s->hash; mp.mask = (unsigned)(s->p.num_buckets - 1); mp.num_buckets = s->p.num_buckets;
  // order keys (4 B/entry) and allocType (1 B/entry) are carved from one scratch block for THIS scene's entry
  // count, so the single memset below clears exactly both
  e->alloc_type = reinterpret_cast<unsigned char *>(e->order_keys) + (size_t)N * 4;
  mp.keys = e->order_keys; mp.alloc_type = e->alloc_type; mp.coords = e->block_coords; mp.vis_type = r->visible_type;
  mp.cnt = s->counters; mp.tile_counts = e->tile_counts;
  // steps along the +-mu segment: ceil(2 * |segment| in blocks) = ceil(mu / (2 * voxel_size)) for a rigid pose
  const int step_bound = (int)ceilf(s->p.mu / (2.0f * s->p.voxel_size)) + 2;
  mp.step_cap = ceil_pow2(step_bound + 1);
  if ((double)W * H * mp.step_cap >= 4294967295.0) {
    set_last_error("image size x ray steps exceeds the 32-bit allocation order key");
    return DSLAM_ERR_UNSUPPORTED;
  }

  const int n_tiles = num_tiles(N);
  // order keys (4N bytes) + allocType (N bytes) = 5N bytes, N % 16 checked below -> whole uint4 stores
  hipLaunchKernelGGL(k_alloc_prep, dim3(1024), dim3(256), 0, e->stream, reinterpret_cast<uint4 *>(e->order_keys),
                     (int)(((size_t)N * 5) / 16), e->tile_counts, n_tiles * 2, r->visible_ids, r->counters, r->visible_type,
                     v->depth_dirty ? v->raw_src : (const short *)nullptr, v->depth, W * H, v->affine_a, v->affine_b);
  v->depth_dirty = false;
  const int pix_blocks = (W * H + 255) / 256;
  hipLaunchKernelGGL(k_mark<0>, dim3(pix_blocks), dim3(256), 0, e->stream, mp);
  hipLaunchKernelGGL(k_mark<1>, dim3(pix_blocks), dim3(256), 0, e->stream, mp);
  if (!only_update_visible_list) {
    hipLaunchKernelGGL(k_commit_apply, dim3(n_tiles), dim3(256), 0, e->stream, e->alloc_type, e->block_coords, N,
                       e->tile_counts, s->hash, s->p.num_buckets, s->alloc_list, s->excess_list, r->visible_type,
                       s->counters, n_tiles);
  }
  const int fin = only_update_visible_list ? 0 : 1;
  VisParams vp;
  memcpy(vp.M.m, M_d, sizeof(float) * 16);
  vp.fx = intr[0]; vp.fy = intr[1]; vp.cx = intr[2]; vp.cy = intr[3]; vp.voxel_size = s->p.voxel_size; vp.W = W; vp.H = H;
  if (s->p.use_swapping)
    hipLaunchKernelGGL(k_visible_count<true>, dim3(n_tiles), dim3(256), 0, e->stream, r->visible_type, s->hash,
                       s->swap_state, N, vp, e->tile_offsets, s->counters, e->tile_counts, n_tiles * 2, fin);
  else
    hipLaunchKernelGGL(k_visible_count<false>, dim3(n_tiles), dim3(256), 0, e->stream, r->visible_type, s->hash,
                       (unsigned char *)nullptr, N, vp, e->tile_offsets, s->counters, e->tile_counts, n_tiles * 2, fin);
  // (tile_offsets holds the visible counts here: the commit request counts in tile_counts are still being read)
  hipLaunchKernelGGL(k_compact_apply_fused, dim3(n_tiles), dim3(256), 0, e->stream, r->visible_type, N, e->tile_offsets,
                     r->visible_ids, r->n_local, &r->counters->no_visible);
  if (s->p.use_swapping) {
    hipLaunchKernelGGL(k_realloc_count, dim3(n_tiles), dim3(256), 0, e->stream, r->visible_type, s->hash, N,
                       e->tile_counts);
    hipLaunchKernelGGL(k_realloc_scan, dim3(1), dim3(1024), 0, e->stream, e->tile_counts, e->tile_offsets, n_tiles,
                       s->counters);
    hipLaunchKernelGGL(k_realloc_apply, dim3(n_tiles), dim3(256), 0, e->stream, r->visible_type, s->hash, N,
                       e->tile_offsets, s->alloc_list, s->counters);
  }
  DSLAM_HIP(hipGetLastError());
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
