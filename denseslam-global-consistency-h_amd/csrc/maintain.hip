// maintain.hip -- voxel decay, sliding window and swap-in/swap-out for gfx950: the memory path that gives the
// reference its "global consistency" behaviour.
//
// Reference call sites: denseMapper->Decay / DecayDefusionPart (InfiniTamDriver.h:274-292,315-331),
// denseMapper->SlideWindow / SlideWindowDefusionPart (InfiniTamDriver.h:294-310), GetDecayedBlockCount (:366-370),
// ITMSwappingEngine::{IntegrateGlobalIntoLocal, SaveToGlobalMemory} (InfiniTamDriver.h:240-242, DenseSlam.h:248-251).
// Semantics: DESIGN.md section 5 (the fork's bodies are not in the reference tree).
//
// Everything is expressed as ordered compactions over the hash table (count -> scan -> apply, ascending entry
// index) followed by one-wave-per-block streaming kernels, so results are deterministic and equal to the
// sequential oracle:
//   candidates  entries whose block carries the ring bit of the list being decayed / popped (or is old enough)
//   decay       wave per candidate block: reset voxels with 0 < w <= maxWeight, wave-ballot "any measured voxel left"
//   release     removal list (ascending entry index): reset block, push slot r-th onto the free stack, clear rings
//   unlink      one lane per affected bucket rewrites the chain once (survivors keep order, first survivor moves
//               into a released head); freed excess slots are flagged and pushed in ascending slot order
//   rebuild     visible list of the render state from visibleType (ordered compaction)
#include "dslam_internal.h"

#pragma clang fp contract(off)

namespace dslam {

// ---- candidate selection ---------------------------------------------------------------------------------------
// MODE 0: block carries bit `bit` of ring `ring` (aged-list decay)
// MODE 1: same, and the bit is cleared; flag only blocks no queued list references any more (sliding-window pop)
// MODE 2: block not seen since `threshold` and not yet swept in this observation epoch (full-sweep decay)
template <int MODE>
__global__ __launch_bounds__(256) void k_select(const HashEntry *__restrict__ hash, int n_entries,
                                                unsigned long long *masks, int words, int ring, int bit,
                                                int *last_seen, int threshold, unsigned char *__restrict__ flags,
                                                int *__restrict__ tile_counts) {
  __shared__ int red[4];
  const int t0 = blockIdx.x * kTileEntries + threadIdx.x * 4;
  int c = 0;
  if (t0 < n_entries) {
    unsigned char f[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int ptr = hash[t0 + k].ptr;
      f[k] = 0;
      if (ptr >= 0) {
        if (MODE == 2) {
          const int ls = last_seen[ptr];
          if (ls >= 0 && ls <= threshold) {
            last_seen[ptr] = -2 - ls;
            f[k] = 1;
          }
        } else {
          unsigned long long *m = masks + ((size_t)ptr * 2) * words;
          unsigned long long &w = m[(size_t)ring * words + (bit >> 6)];
          const unsigned long long b = 1ull << (bit & 63);
          if (w & b) {
            if (MODE == 0) {
              f[k] = 1;
            } else {
              w &= ~b;
              unsigned long long any = 0;
              for (int i = 0; i < 2 * words; i++) any |= m[i];
              f[k] = any ? 0 : 1;
            }
          }
        }
      }
      c += f[k];
    }
    *reinterpret_cast<uchar4 *>(flags + t0) = make_uchar4(f[0], f[1], f[2], f[3]);
  }
  int tot;
  block_excl_scan<4>(c, red, tot);
  if (threadIdx.x == 0) tile_counts[blockIdx.x] = tot;
}

// ---- decay: one wavefront per candidate block ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_decay_blocks(const int *__restrict__ cand, const int *count_ptr,
                                                      const HashEntry *__restrict__ hash, uint4 *voxels16,
                                                      int max_weight, unsigned char *remove_flags, int mark_empty) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256 + threadIdx.x) >> 6));
  const int n_waves = gridDim.x * 4;
  const int n = *count_ptr;
  for (int i = wave; i < n; i += n_waves) {
    const int t = cand[i];
    const int ptr = hash[t].ptr;
    uint4 *blk = voxels16 + (size_t)ptr * (kBlock3 / 2);
    bool measured = false;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      uint4 v = blk[j * 64 + lane];
      bool ch = false;
      const unsigned w0 = (v.x >> 16) & 0xffu, w1 = (v.z >> 16) & 0xffu;
      if (w0 > 0 && (int)w0 <= max_weight) { v.x = kEmptyVoxelLo; v.y = kEmptyVoxelHi; ch = true; }
      if (w1 > 0 && (int)w1 <= max_weight) { v.z = kEmptyVoxelLo; v.w = kEmptyVoxelHi; ch = true; }
      measured |= (((v.x >> 16) & 0xffu) > 0) || (((v.z >> 16) & 0xffu) > 0);
      if (ch) blk[j * 64 + lane] = v;
    }
    const bool any = __ballot(measured) != 0ull;
    if (!any && mark_empty && lane == 0) remove_flags[t] = 1;
  }
}

// ---- release: removal list -> pool -----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_release_blocks(const int *__restrict__ rem, const SceneCounters *cnt,
                                                        const HashEntry *__restrict__ hash, uint4 *voxels16,
                                                        int *alloc_list, unsigned long long *masks, int *last_seen,
                                                        int words) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256 + threadIdx.x) >> 6));
  const int n_waves = gridDim.x * 4;
  const int n = cnt->remove_count;
  const int base = cnt->last_free;
  const uint4 empty2 = make_uint4(kEmptyVoxelLo, kEmptyVoxelHi, kEmptyVoxelLo, kEmptyVoxelHi);
  for (int r = wave; r < n; r += n_waves) {
    const int ptr = hash[rem[r]].ptr;
    uint4 *blk = voxels16 + (size_t)ptr * (kBlock3 / 2);
#pragma unroll
    for (int j = 0; j < 4; j++) blk[j * 64 + lane] = empty2;
    if (lane < 2 * words) masks[(size_t)ptr * 2 * words + lane] = 0ull;
    if (lane == 0) {
      alloc_list[base + 1 + r] = ptr;
      last_seen[ptr] = -1;
    }
  }
}

// one lane per removal: am I the first released entry of my bucket chain (chain still unmodified)?
__global__ __launch_bounds__(256) void k_find_leaders(const int *__restrict__ rem, const SceneCounters *cnt,
                                                      const HashEntry *__restrict__ hash, int num_buckets,
                                                      unsigned mask, const unsigned char *__restrict__ remove_flags,
                                                      int *__restrict__ leader_bucket) {
  const int n = cnt->remove_count;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
    const int t = rem[r];
    int bucket = t;
    if (t >= num_buckets) {
      const HashEntry e = load_entry(hash, t);
      bucket = hash_index(e.pos[0], e.pos[1], e.pos[2], mask);
    }
    int c = bucket, first = -1;
    while (c >= 0) {
      if (remove_flags[c]) { first = c; break; }
      const int off = hash[c].offset;
      c = (off >= 1) ? num_buckets + off - 1 : -1;
    }
    leader_bucket[r] = (first == t) ? bucket : -1;
  }
}

// leaders rewrite their bucket chain once (DESIGN.md "batch release")
__global__ __launch_bounds__(256) void k_unlink(const int *__restrict__ leader_bucket, const SceneCounters *cnt,
                                                HashEntry *hash, int num_buckets,
                                                const unsigned char *__restrict__ remove_flags,
                                                unsigned char *freed_flags, unsigned char *vis_type) {
  const int n = cnt->remove_count;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
    const int head = leader_bucket[r];
    if (head < 0) continue;
    int c = head, prev = -1;
    while (c >= 0) {
      const HashEntry e = load_entry(hash, c);
      const int next = (e.offset >= 1) ? num_buckets + e.offset - 1 : -1;
      if (remove_flags[c]) {
        if (c != head) freed_flags[c - num_buckets] = 1;
        store_entry(hash, c, 0, 0, 0, 0, -2);
        if (vis_type) vis_type[c] = 0;
      } else {
        int cur = c;
        if (prev == -1) {
          if (c != head) {  // first survivor moves into the released bucket head
            store_entry(hash, head, e.pos[0], e.pos[1], e.pos[2], e.offset, e.ptr);
            if (vis_type) { vis_type[head] = vis_type[c]; vis_type[c] = 0; }
            store_entry(hash, c, 0, 0, 0, 0, -2);
            freed_flags[c - num_buckets] = 1;
            cur = head;
          }
        } else {
          hash[prev].offset = (c - num_buckets) + 1;
        }
        prev = cur;
      }
      c = next;
    }
    if (prev >= 0) hash[prev].offset = 0;
  }
}

// push freed excess slots in ascending slot order
__global__ __launch_bounds__(256) void k_push_freed(const unsigned char *__restrict__ freed_flags, int n_excess,
                                                    const int *__restrict__ tile_offsets, int *excess_list,
                                                    const SceneCounters *cnt) {
  __shared__ int red[4];
  const int t0 = blockIdx.x * kTileEntries + threadIdx.x * 4;
  unsigned char f[4] = {0, 0, 0, 0};
  if (t0 < n_excess) {
    const uchar4 v = *reinterpret_cast<const uchar4 *>(freed_flags + t0);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  }
  const int c = (f[0] > 0) + (f[1] > 0) + (f[2] > 0) + (f[3] > 0);
  int tot;
  int r = block_excl_scan<4>(c, red, tot);
  if (tot == 0) return;
  r += tile_offsets[blockIdx.x] + cnt->last_free_ex + 1;
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (f[k] > 0) excess_list[r++] = t0 + k;
}

__global__ void k_finalize_removal(SceneCounters *cnt, int count_as_slid) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int n = cnt->remove_count;
  cnt->last_free += n;
  cnt->last_free_ex += cnt->freed_excess;
  if (count_as_slid) cnt->slid_blocks += n; else cnt->decayed_blocks += n;
  cnt->remove_count = 0;
  cnt->freed_excess = 0;
}

// scratch carving (engine->list_c holds >= 4*N bytes): candidate flags, removal flags, freed-excess flags
struct MaintScratch {
  unsigned char *cand_flags, *rem_flags, *freed_flags;
  int *cand_list, *rem_list, *leaders;
};
static MaintScratch carve(dslam_engine *e, int N) {
  MaintScratch m;
  unsigned char *b = reinterpret_cast<unsigned char *>(e->list_c);
  m.cand_flags = b;
  m.rem_flags = b + N;
  m.freed_flags = b + 2 * (size_t)N;
  m.cand_list = e->list_a;
  m.rem_list = e->list_b;
  m.leaders = e->list_d;
  return m;
}

static int rebuild_visible_list(dslam_engine *e, dslam_render_state *r) {
  const int N = r->n_entries, n_tiles = num_tiles(N);
  // (each apply workgroup sums the preceding tile counts itself: no scan launch in between)
  hipLaunchKernelGGL(k_flag_count, dim3(n_tiles), dim3(256), 0, e->stream, r->visible_type, N, e->tile_counts);
  hipLaunchKernelGGL(k_compact_apply_fused, dim3(n_tiles), dim3(256), 0, e->stream, r->visible_type, N, e->tile_counts,
                     r->visible_ids, r->n_local, &r->counters->no_visible);
  DSLAM_HIP(hipGetLastError());
  return DSLAM_OK;
}

// removal flags (entry-indexed) -> removal list -> release + unlink + free-list pushes + visible-list rebuild
static int release_flagged(dslam_engine *e, dslam_scene *s, dslam_render_state *r, const MaintScratch &m,
                           const unsigned char *rem_flags, int count_as_slid) {
  const int N = s->n_entries, n_tiles = num_tiles(N);
  const int x_tiles = num_tiles(s->p.num_excess);
  hipLaunchKernelGGL(k_flag_count, dim3(n_tiles), dim3(256), 0, e->stream, rem_flags, N, e->tile_counts);
  hipLaunchKernelGGL(k_compact_apply_fused, dim3(n_tiles), dim3(256), 0, e->stream, rem_flags, N, e->tile_counts, m.rem_list,
                     s->p.num_local_blocks, &s->counters->remove_count);
  hipLaunchKernelGGL(k_release_blocks, dim3(1024), dim3(256), 0, e->stream, m.rem_list, s->counters, s->hash,
                     reinterpret_cast<uint4 *>(s->voxels), s->alloc_list, s->masks, s->last_seen, s->history_words);
  hipLaunchKernelGGL(k_find_leaders, dim3(256), dim3(256), 0, e->stream, m.rem_list, s->counters, s->hash,
                     s->p.num_buckets, (unsigned)(s->p.num_buckets - 1), rem_flags, m.leaders);
  DSLAM_HIP(hipMemsetAsync(m.freed_flags, 0, (size_t)x_tiles * kTileEntries, e->stream));
  hipLaunchKernelGGL(k_unlink, dim3(256), dim3(256), 0, e->stream, m.leaders, s->counters, s->hash, s->p.num_buckets,
                     rem_flags, m.freed_flags, r ? r->visible_type : (unsigned char *)nullptr);
  hipLaunchKernelGGL(k_flag_count, dim3(x_tiles), dim3(256), 0, e->stream, m.freed_flags, s->p.num_excess,
                     e->tile_counts);
  hipLaunchKernelGGL(k_scan_count, dim3(1), dim3(1024), 0, e->stream, e->tile_counts, e->tile_offsets, x_tiles,
                     &s->counters->freed_excess, s->p.num_excess);
  hipLaunchKernelGGL(k_push_freed, dim3(x_tiles), dim3(256), 0, e->stream, m.freed_flags, s->p.num_excess,
                     e->tile_offsets, s->excess_list, s->counters);
  hipLaunchKernelGGL(k_finalize_removal, dim3(1), dim3(64), 0, e->stream, s->counters, count_as_slid);
  DSLAM_HIP(hipGetLastError());
  if (r) return rebuild_visible_list(e, r);
  return DSLAM_OK;
}

static int decay_candidates(dslam_engine *e, dslam_scene *s, dslam_render_state *r, const MaintScratch &m,
                            int max_weight) {
  // cand_flags / tile_counts hold the selection; turn it into the candidate list, decay, release empties
  const int N = s->n_entries, n_tiles = num_tiles(N);
  hipLaunchKernelGGL(k_compact_apply_fused, dim3(n_tiles), dim3(256), 0, e->stream, m.cand_flags, N, e->tile_counts,
                     m.cand_list, s->p.num_local_blocks, &s->counters->swap_count);  // swap_count doubles as candidate count
  DSLAM_HIP(hipMemsetAsync(m.rem_flags, 0, N, e->stream));
  hipLaunchKernelGGL(k_decay_blocks, dim3(1024), dim3(256), 0, e->stream, m.cand_list, &s->counters->swap_count, s->hash,
                     reinterpret_cast<uint4 *>(s->voxels), max_weight, m.rem_flags, s->p.use_swapping ? 0 : 1);
  DSLAM_HIP(hipGetLastError());
  if (s->p.use_swapping) return DSLAM_OK;  // entries of a swapping scene are never unlinked (ITMGlobalCache keys)
  return release_flagged(e, s, r, m, m.rem_flags, 0);
}

int launch_decay(dslam_engine *e, dslam_scene *s, dslam_render_state *r, int max_weight, int min_age, int force_all,
                 int q) {
  DSLAM_REQUIRE(!r || r->n_entries == s->n_entries, "render state was created for a different scene size");
  const int N = s->n_entries, n_tiles = num_tiles(N);
  int rc = ensure_scratch(e, N, s->p.num_local_blocks);
  if (rc) return rc;
  const MaintScratch m = carve(e, N);
  if (!force_all) {
    const int bits = 64 * s->history_words;
    const int newest = s->ring_next[q] - 1;
    int k = s->decay_cursor[q] > s->ring_head[q] ? s->decay_cursor[q] : s->ring_head[q];
    for (; k <= newest - min_age; k++) {
      hipLaunchKernelGGL(k_select<0>, dim3(n_tiles), dim3(256), 0, e->stream, s->hash, N, s->masks, s->history_words, q,
                         k % bits, s->last_seen, 0, m.cand_flags, e->tile_counts);
      if ((rc = decay_candidates(e, s, r, m, max_weight))) return rc;
    }
    if (k > s->decay_cursor[q]) s->decay_cursor[q] = k;
  } else {
    const int threshold = (s->frame_counter - 1) - min_age;
    hipLaunchKernelGGL(k_select<2>, dim3(n_tiles), dim3(256), 0, e->stream, s->hash, N, s->masks, s->history_words, q, 0,
                       s->last_seen, threshold, m.cand_flags, e->tile_counts);
    if ((rc = decay_candidates(e, s, r, m, max_weight))) return rc;
  }
  return DSLAM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// swapping (ITMSwappingEngine + ITMGlobalCache).  Host involvement (the global cache lives in host
// memory) makes these calls synchronous.
// ---------------------------------------------------------------------------------------------------------------
// MODE 0: swap state == 1 (needs the host copy merged)     -- IntegrateGlobalIntoLocal
// MODE 1: resident with state 0 (never visible since allocation) -- flush promotion
// MODE 2: state == 2, resident, not visible (or any visibility) -- SaveToGlobalMemory
template <int MODE>
__global__ __launch_bounds__(256) void k_swap_select(const HashEntry *__restrict__ hash, int n_entries,
                                                     const unsigned char *__restrict__ swap_state,
                                                     const unsigned char *__restrict__ vis_type,
                                                     unsigned char *__restrict__ flags, int *__restrict__ tile_counts) {
  __shared__ int red[4];
  const int t0 = blockIdx.x * kTileEntries + threadIdx.x * 4;
  int c = 0;
  if (t0 < n_entries) {
    unsigned char f[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int t = t0 + k;
      const unsigned char st = swap_state[t];
      bool sel;
      if (MODE == 0) sel = st == 1;
      else if (MODE == 1) sel = st == 0 && hash[t].ptr >= 0;
      else sel = st == 2 && hash[t].ptr >= 0 && (vis_type == nullptr || vis_type[t] == 0);
      f[k] = sel ? 1 : 0;
      c += f[k];
    }
    *reinterpret_cast<uchar4 *>(flags + t0) = make_uchar4(f[0], f[1], f[2], f[3]);
  }
  int tot;
  block_excl_scan<4>(c, red, tot);
  if (threadIdx.x == 0) tile_counts[blockIdx.x] = tot;
}

// CombineVoxelInformation: merge the host copy (src) into the resident voxel (dst)
__device__ __forceinline__ void combine_voxel(unsigned slo, unsigned shi, unsigned &dlo, unsigned &dhi, int maxW) {
  {
    int newW = (int)((dlo >> 16) & 0xffu);
    const int oldW = (int)((slo >> 16) & 0xffu);
    if (oldW != 0) {
      float newF = sdf_to_float((short)(dlo & 0xffffu));
      const float oldF = sdf_to_float((short)(slo & 0xffffu));
      newF = (float)oldW * oldF + (float)newW * newF;
      newW = oldW + newW;
      newF /= (float)newW;
      newW = newW < maxW ? newW : maxW;
      dlo = (dlo & 0xff000000u) | ((unsigned)newW << 16) | (unsigned)(unsigned short)float_to_sdf(newF);
    }
  }
  {
    const int newW = (int)((dhi >> 16) & 0xffu), oldW = (int)((shi >> 16) & 0xffu);
    if (oldW != 0) {
      const int sumW = oldW + newW;
      const unsigned dc[3] = {dlo >> 24, dhi & 0xffu, (dhi >> 8) & 0xffu};
      const unsigned sc[3] = {slo >> 24, shi & 0xffu, (shi >> 8) & 0xffu};
      unsigned nc[3];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        float v = (float)dc[k] / 255.0f;
        const float oc = (float)sc[k] / 255.0f;
        v = oc * (float)oldW + v * (float)newW;
        v /= (float)sumW;
        nc[k] = (unsigned)(unsigned char)(v * 255.0f);
      }
      const unsigned w = (unsigned)(sumW < maxW ? sumW : maxW);
      dlo = (dlo & 0x00ffffffu) | (nc[0] << 24);
      dhi = (dhi & 0xff000000u) | nc[1] | (nc[2] << 8) | (w << 16);
    }
  }
}

// address of a stored block in the page-locked host slabs
__device__ __forceinline__ uint4 *stored_block(uint4 *const *slabs, int slot) {
  return slabs[slot >> kSlabShift] + (size_t)(slot & (kSlabBlocks - 1)) * (kBlock3 / 2);
}

// merge the host copies of ids[0..n) into their (re-allocated) blocks; the stored blocks are read straight from the
// host slabs over PCIe (slot < 0: the entry has no host copy)
__global__ __launch_bounds__(256) void k_swap_merge(const int *__restrict__ ids, const int *__restrict__ slots,
                                                    int n, const HashEntry *__restrict__ hash, uint4 *voxels16,
                                                    uint4 *const *__restrict__ slabs, unsigned char *swap_state,
                                                    int maxW) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256 + threadIdx.x) >> 6));
  const int n_waves = gridDim.x * 4;
  for (int i = wave; i < n; i += n_waves) {
    const int t = ids[i];
    const int ptr = hash[t].ptr;
    const int slot = slots[i];
    if (slot >= 0 && ptr >= 0) {
      uint4 *blk = voxels16 + (size_t)ptr * (kBlock3 / 2);
      const uint4 *src = stored_block(slabs, slot);
      uint4 sv[4];
#pragma unroll
      for (int j = 0; j < 4; j++) sv[j] = src[j * 64 + lane];  // all four PCIe reads in flight
#pragma unroll
      for (int j = 0; j < 4; j++) {
        uint4 d = blk[j * 64 + lane];
        combine_voxel(sv[j].x, sv[j].y, d.x, d.y, maxW);
        combine_voxel(sv[j].z, sv[j].w, d.z, d.w, maxW);
        blk[j * 64 + lane] = d;
      }
    }
    if (lane == 0) swap_state[t] = 2;
  }
}

// write selected blocks to their host slots (over PCIe, straight from the kernel), reset them, give their voxel-block
// slots back, mark the entries swapped out
__global__ __launch_bounds__(256) void k_swap_pack(const int *__restrict__ ids, const int *__restrict__ slots, int n,
                                                   HashEntry *hash, uint4 *voxels16, uint4 *const *__restrict__ slabs,
                                                   int *alloc_list, unsigned long long *masks,
                                                   int *last_seen, int words, unsigned char *swap_state,
                                                   unsigned char *vis_type, SceneCounters *cnt) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256 + threadIdx.x) >> 6));
  const int n_waves = gridDim.x * 4;
  const int base = cnt->last_free;
  const uint4 empty2 = make_uint4(kEmptyVoxelLo, kEmptyVoxelHi, kEmptyVoxelLo, kEmptyVoxelHi);
  for (int i = wave; i < n; i += n_waves) {
    const int t = ids[i];
    const int ptr = hash[t].ptr;
    uint4 *blk = voxels16 + (size_t)ptr * (kBlock3 / 2);
    uint4 *dst = stored_block(slabs, slots[i]);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      dst[j * 64 + lane] = blk[j * 64 + lane];
      blk[j * 64 + lane] = empty2;
    }
    if (lane < 2 * words) masks[(size_t)ptr * 2 * words + lane] = 0ull;
    if (lane == 0) {
      alloc_list[base + 1 + i] = ptr;
      last_seen[ptr] = -1;
      hash[t].ptr = -1;
      swap_state[t] = 0;
      if (vis_type) vis_type[t] = 0;
    }
  }
}

__global__ void k_add_last_free(SceneCounters *cnt, int n, int add_slid) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    cnt->last_free += n;
    if (add_slid) cnt->slid_blocks += n;
  }
}

// select (ordered, capped at the transfer-buffer size) and bring ids + count to the host
template <int MODE>
static int swap_select_to_host(dslam_engine *e, dslam_scene *s, const unsigned char *vis_type, const MaintScratch &m,
                               int *out_count) {
  const int N = s->n_entries, n_tiles = num_tiles(N);
  hipLaunchKernelGGL(k_swap_select<MODE>, dim3(n_tiles), dim3(256), 0, e->stream, s->hash, N, s->swap_state, vis_type,
                     m.cand_flags, e->tile_counts);
  hipLaunchKernelGGL(k_scan_count, dim3(1), dim3(1024), 0, e->stream, e->tile_counts, e->tile_offsets, n_tiles,
                     &s->counters->swap_count, kTransferBlocks);
  hipLaunchKernelGGL(k_compact_apply, dim3(n_tiles), dim3(256), 0, e->stream, m.cand_flags, N, e->tile_offsets,
                     m.cand_list, kTransferBlocks);
  DSLAM_HIP(hipGetLastError());
  // count and ids travel together (16 KiB at most): one wait instead of two
  int *host_count = reinterpret_cast<int *>(e->pinned) + 48;
  DSLAM_HIP(hipMemcpyAsync(host_count, &s->counters->swap_count, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  DSLAM_HIP(hipMemcpyAsync(s->transfer_ids_host, m.cand_list, (size_t)kTransferBlocks * sizeof(int), hipMemcpyDeviceToHost,
                           e->stream));
  DSLAM_HIP(hipStreamSynchronize(e->stream));
  *out_count = *host_count;
  return DSLAM_OK;
}

// one more page-locked slab for the host store; its pointer goes to the device table the kernels index
static int add_slab(dslam_engine *e, dslam_scene *s) {
  if ((int)s->slabs.size() >= kMaxSlabs) { set_last_error("host global cache: slab table full"); return DSLAM_ERR_INVALID; }
  uint4 *slab = nullptr;
  DSLAM_HIP(hipHostMalloc((void **)&slab, (size_t)kSlabBlocks * kBlock3 * sizeof(uint2), hipHostMallocDefault));
  s->slabs.push_back(slab);
  DSLAM_HIP(hipMemcpyAsync(s->slab_ptrs_dev + (s->slabs.size() - 1), &s->slabs.back(), sizeof(uint4 *), hipMemcpyHostToDevice,
                           e->stream));
  DSLAM_HIP(hipStreamSynchronize(e->stream));  // (the source is a vector element)
  return DSLAM_OK;
}

// slots of ids[0..n) into the second half of the pinned id buffer and on to the device; `assign`: entries without a
// host copy get the next free slot (swap-out), otherwise they keep -1 (swap-in of an entry that was never stored)
static int batch_slots_to_device(dslam_engine *e, dslam_scene *s, int n, bool assign, int *slots_dev) {
  const int *ids = s->transfer_ids_host;
  int *slots = s->transfer_ids_host + kTransferBlocks;
  for (int i = 0; i < n; i++) {
    int slot = s->slot_host[ids[i]];
    if (slot < 0 && assign) {
      slot = s->next_slot++;
      while ((slot >> kSlabShift) >= (int)s->slabs.size()) {
        int rc = add_slab(e, s);
        if (rc) return rc;
      }
      s->slot_host[ids[i]] = slot;
    }
    slots[i] = slot;
  }
  DSLAM_HIP(hipMemcpyAsync(slots_dev, slots, (size_t)n * sizeof(int), hipMemcpyHostToDevice, e->stream));
  return DSLAM_OK;
}

// merge the host copies of ids[0..n) (ids in m.cand_list and in the pinned id buffer) into the map
static int merge_from_host(dslam_engine *e, dslam_scene *s, const MaintScratch &m, int n) {
  int *slots_dev = m.rem_list;  // free int scratch
  int rc = batch_slots_to_device(e, s, n, false, slots_dev);
  if (rc) return rc;
  hipLaunchKernelGGL(k_swap_merge, dim3(1024), dim3(256), 0, e->stream, m.cand_list, slots_dev, n, s->hash,
                     reinterpret_cast<uint4 *>(s->voxels), s->slab_ptrs_dev, s->swap_state, s->p.max_w);
  DSLAM_HIP(hipGetLastError());
  // no wait here: the pinned id / slot buffer is next written by the host only after the next selection's wait, and
  // by the device only in stream order
  return DSLAM_OK;
}

// write ids[0..n) (in m.cand_list and in the pinned id buffer) to the host store and release their voxel blocks
static int pack_to_host(dslam_engine *e, dslam_scene *s, unsigned char *vis_type, const MaintScratch &m, int n,
                        int add_slid) {
  int *slots_dev = m.rem_list;
  int rc = batch_slots_to_device(e, s, n, true, slots_dev);
  if (rc) return rc;
  hipLaunchKernelGGL(k_swap_pack, dim3(1024), dim3(256), 0, e->stream, m.cand_list, slots_dev, n, s->hash,
                     reinterpret_cast<uint4 *>(s->voxels), s->slab_ptrs_dev, s->alloc_list, s->masks,
                     s->last_seen, s->history_words, s->swap_state, vis_type, s->counters);
  hipLaunchKernelGGL(k_add_last_free, dim3(1), dim3(64), 0, e->stream, s->counters, n, add_slid);
  DSLAM_HIP(hipGetLastError());
  return DSLAM_OK;  // (no wait: see merge_from_host; readers of the host store synchronise first)
}

int launch_swap_in(dslam_engine *e, dslam_scene *s, dslam_render_state *) {
  int rc = ensure_scratch(e, s->n_entries, s->p.num_local_blocks);
  if (rc) return rc;
  const MaintScratch m = carve(e, s->n_entries);
  int n = 0;
  if ((rc = swap_select_to_host<0>(e, s, nullptr, m, &n))) return rc;
  if (n > 0 && (rc = merge_from_host(e, s, m, n))) return rc;
  s->last_swapped_in = n;
  return DSLAM_OK;
}

int launch_swap_out(dslam_engine *e, dslam_scene *s, dslam_render_state *r, bool ignore_visibility) {
  int rc = ensure_scratch(e, s->n_entries, s->p.num_local_blocks);
  if (rc) return rc;
  const MaintScratch m = carve(e, s->n_entries);
  int n = 0;
  if ((rc = swap_select_to_host<2>(e, s, ignore_visibility ? nullptr : r->visible_type, m, &n))) return rc;
  if (n > 0 && (rc = pack_to_host(e, s, nullptr, m, n, 0))) return rc;
  s->last_swapped_out = n;
  return DSLAM_OK;
}

// Hansry's SaveToGlobalMemory(scene): merge everything pending, promote never-visible resident blocks, flush all
int launch_save_to_global(dslam_engine *e, dslam_scene *s) {
  int rc = ensure_scratch(e, s->n_entries, s->p.num_local_blocks);
  if (rc) return rc;
  const MaintScratch m = carve(e, s->n_entries);
  int n = 0;
  while (true) {
    if ((rc = swap_select_to_host<0>(e, s, nullptr, m, &n))) return rc;
    if (n == 0) break;
    if ((rc = merge_from_host(e, s, m, n))) return rc;
  }
  s->last_swapped_in = 0;
  while (true) {
    if ((rc = swap_select_to_host<1>(e, s, nullptr, m, &n))) return rc;
    if (n == 0) break;
    if ((rc = merge_from_host(e, s, m, n))) return rc;
  }
  int total = 0;
  while (true) {
    if ((rc = swap_select_to_host<2>(e, s, nullptr, m, &n))) return rc;
    if (n == 0) break;
    if ((rc = pack_to_host(e, s, nullptr, m, n, 0))) return rc;
    total += n;
  }
  s->last_swapped_out = total;
  return DSLAM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// sliding window: pop the oldest list of ring q
// ---------------------------------------------------------------------------------------------------------------
// entries flagged for leaving (swapping scene) whose host copy is not merged yet: state != 2
__global__ __launch_bounds__(256) void k_slide_split(const unsigned char *__restrict__ leave_flags, int n_entries,
                                                     const unsigned char *__restrict__ swap_state,
                                                     unsigned char *__restrict__ need_merge_flags,
                                                     int *__restrict__ tile_counts) {
  __shared__ int red[4];
  const int t0 = blockIdx.x * kTileEntries + threadIdx.x * 4;
  int c = 0;
  if (t0 < n_entries) {
    unsigned char f[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      f[k] = (leave_flags[t0 + k] && swap_state[t0 + k] != 2) ? 1 : 0;
      c += f[k];
    }
    *reinterpret_cast<uchar4 *>(need_merge_flags + t0) = make_uchar4(f[0], f[1], f[2], f[3]);
  }
  int tot;
  block_excl_scan<4>(c, red, tot);
  if (threadIdx.x == 0) tile_counts[blockIdx.x] = tot;
}

// compaction of `flags` skipping the first `skip` hits, at most kTransferBlocks outputs (batched host transfers)
__global__ __launch_bounds__(256) void k_compact_window(const unsigned char *__restrict__ flags, int n_entries,
                                                        const int *__restrict__ tile_offsets, int *__restrict__ out,
                                                        int skip, int capacity) {
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
  r += tile_offsets[blockIdx.x] - skip;
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (f[k] > 0) {
      if (r >= 0 && r < capacity) out[r] = t0 + k;
      r++;
    }
}

static int flags_to_host_batches(dslam_engine *e, dslam_scene *s, const unsigned char *flags, const MaintScratch &m,
                                 int *total_out) {
  const int N = s->n_entries, n_tiles = num_tiles(N);
  hipLaunchKernelGGL(k_flag_count, dim3(n_tiles), dim3(256), 0, e->stream, flags, N, e->tile_counts);
  hipLaunchKernelGGL(k_scan_count, dim3(1), dim3(1024), 0, e->stream, e->tile_counts, e->tile_offsets, n_tiles,
                     &s->counters->swap_count, s->p.num_local_blocks);
  DSLAM_HIP(hipGetLastError());
  int *host_count = reinterpret_cast<int *>(e->pinned) + 48;
  DSLAM_HIP(hipMemcpyAsync(host_count, &s->counters->swap_count, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  DSLAM_HIP(hipStreamSynchronize(e->stream));
  *total_out = *host_count;
  return DSLAM_OK;
}

static int batch_ids_to_host(dslam_engine *e, dslam_scene *s, const unsigned char *flags, const MaintScratch &m, int skip,
                             int n) {
  const int N = s->n_entries, n_tiles = num_tiles(N);
  hipLaunchKernelGGL(k_compact_window, dim3(n_tiles), dim3(256), 0, e->stream, flags, N, e->tile_offsets, m.cand_list, skip,
                     kTransferBlocks);
  DSLAM_HIP(hipGetLastError());
  DSLAM_HIP(hipMemcpyAsync(s->transfer_ids_host, m.cand_list, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, e->stream));
  DSLAM_HIP(hipStreamSynchronize(e->stream));
  return DSLAM_OK;
}

int launch_slide_pop(dslam_engine *e, dslam_scene *s, dslam_render_state *r, int q) {
  DSLAM_REQUIRE(!r || r->n_entries == s->n_entries, "render state was created for a different scene size");
  const int N = s->n_entries, n_tiles = num_tiles(N);
  int rc = ensure_scratch(e, N, s->p.num_local_blocks);
  if (rc) return rc;
  const MaintScratch m = carve(e, N);
  const int bits = 64 * s->history_words;
  const int bit = (s->ring_head[q]++) % bits;
  if (s->decay_cursor[q] < s->ring_head[q]) s->decay_cursor[q] = s->ring_head[q];
  // rem_flags[t] = 1 for blocks whose rings are empty after clearing this list's bit
  hipLaunchKernelGGL(k_select<1>, dim3(n_tiles), dim3(256), 0, e->stream, s->hash, N, s->masks, s->history_words, q, bit,
                     s->last_seen, 0, m.rem_flags, e->tile_counts);
  DSLAM_HIP(hipGetLastError());
  if (!s->p.use_swapping) return release_flagged(e, s, r, m, m.rem_flags, 1);

  // scene with swapping: the blocks move to the host store, their entries stay (ptr = -1)
  int total = 0;
  // (1) leaving blocks whose host copy was never merged (state != 2): merge it first
  hipLaunchKernelGGL(k_slide_split, dim3(n_tiles), dim3(256), 0, e->stream, m.rem_flags, N, s->swap_state, m.cand_flags,
                     e->tile_counts);
  hipLaunchKernelGGL(k_scan_count, dim3(1), dim3(1024), 0, e->stream, e->tile_counts, e->tile_offsets, n_tiles,
                     &s->counters->swap_count, s->p.num_local_blocks);
  DSLAM_HIP(hipGetLastError());
  int *host_count = reinterpret_cast<int *>(e->pinned) + 48;
  DSLAM_HIP(hipMemcpyAsync(host_count, &s->counters->swap_count, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  DSLAM_HIP(hipStreamSynchronize(e->stream));
  total = *host_count;
  for (int done = 0; done < total; done += kTransferBlocks) {
    const int n = (total - done) < kTransferBlocks ? (total - done) : kTransferBlocks;
    // tile_offsets still hold the scan of the need-merge flags
    if ((rc = batch_ids_to_host(e, s, m.cand_flags, m, done, n))) return rc;
    if ((rc = merge_from_host(e, s, m, n))) return rc;
    // tile scan of cand_flags is unchanged by the merge; keep going
  }
  // (2) pack every leaving block to the host in batches
  if ((rc = flags_to_host_batches(e, s, m.rem_flags, m, &total))) return rc;
  for (int done = 0; done < total; done += kTransferBlocks) {
    const int n = (total - done) < kTransferBlocks ? (total - done) : kTransferBlocks;
    if ((rc = batch_ids_to_host(e, s, m.rem_flags, m, done, n))) return rc;
    if ((rc = pack_to_host(e, s, r ? r->visible_type : nullptr, m, n, 1))) return rc;
  }
  if (r && total > 0) return rebuild_visible_list(e, r);
  return DSLAM_OK;
}

}  // namespace dslam
