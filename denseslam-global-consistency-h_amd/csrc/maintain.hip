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
//   decay       wave per candidate block: reset voxels with 0 < w <= maxWeight, wave-ballot "any measured voxel left";
//               the removal list is compacted from the CANDIDATE list (a few hundred to a few thousand items), not from
//               the table
//   release     removal list (ascending entry index): reset block, push slot r-th onto the free stack, clear rings;
//               the same launch finds, for every removal, whether it is the first one of its bucket chain
//   unlink      one lane per affected bucket rewrites the chain once (survivors keep order, first survivor moves
//               into a released head); freed excess slots are flagged
//   push        the freed excess slots go back in ascending slot order (single-pass ordered compaction with in-launch
//               look-back), the same launch folds the pass into the pool counters
//   rebuild     visible list of the render state from visibleType -- only if the pass took an entry out of it
// Every flag array of this pipeline is all zero between passes: the kernel that consumes a flag clears it, so no pass
// starts with a memset.  Round 1 spent 16 (decay) + 13 (window pop) launches per keyframe on this path; now 8 + 6, all of
// which return at once when their device-side count is zero.
#include "dslam_bits.h"

#pragma clang fp contract(off)

namespace dslam {

// ---- candidate selection ---------------------------------------------------------------------------------------
// Selections over the scene's alloc_bits (entries with a resident block), dslam_bits.h: ONE launch gives the ascending
// list and its length; only entries that hold a block are looked at (round 2: two launches over all 1.18 M entries).
//   SelDecayAged   block carries bit `bit` of ring `ring` (aged-list decay)
//   SelSlidePop    same, and the bit is cleared; selected only if no queued list references the block any more
//   SelDecaySweep  block not seen since `threshold` and not yet swept in this observation epoch (full-sweep decay)
// (the payload of these three is the entry's block slot: the kernel requests it for all of a lane's candidates before the
// first test follows it to the block's history words / age)
struct SelDecayAged {
  typedef int Payload;
  DSLAM_SEL_NO_STAGE
  const HashEntry *hash;
  const unsigned long long *masks;
  int words, ring, bit;
  __device__ int load(int t) const { return hash[t].ptr; }
  __device__ bool test(int, const int &ptr) const {
    if (ptr < 0) return false;
    return (masks[((size_t)ptr * 2 + ring) * words + (bit >> 6)] >> (bit & 63)) & 1ull;
  }
  __device__ void prologue() const {}
  __device__ int emit(int, int, bool, const int &) const { return 0; }
  __device__ void finish(int) const {}
};
struct SelSlidePop {
  typedef int Payload;
  DSLAM_SEL_NO_STAGE
  const HashEntry *hash;
  unsigned long long *masks;
  int words, ring, bit;
  unsigned char *flags;   // (optional) byte flag per selected entry: the release pipeline's removal flags
  __device__ int load(int t) const { return hash[t].ptr; }
  __device__ bool test(int, const int &ptr) const {
    if (ptr < 0) return false;
    unsigned long long *m = masks + ((size_t)ptr * 2) * words;
    unsigned long long &w = m[(size_t)ring * words + (bit >> 6)];
    const unsigned long long b = 1ull << (bit & 63);
    if (!(w & b)) return false;
    w &= ~b;
    unsigned long long any = 0;
    for (int i = 0; i < 2 * words; i++) any |= m[i];
    return any == 0;
  }
  __device__ void prologue() const {}
  __device__ int emit(int t, int, bool, const int &) const { if (flags) flags[t] = 1; return 0; }
  __device__ void finish(int) const {}
};
struct SelDecaySweep {
  typedef int Payload;
  DSLAM_SEL_NO_STAGE
  const HashEntry *hash;
  int *last_seen;
  int threshold;
  __device__ int load(int t) const { return hash[t].ptr; }
  __device__ bool test(int, const int &ptr) const {
    if (ptr < 0) return false;
    const int ls = last_seen[ptr];
    if (ls >= 0 && ls <= threshold) { last_seen[ptr] = -2 - ls; return true; }
    return false;
  }
  __device__ void prologue() const {}
  __device__ int emit(int, int, bool, const int &) const { return 0; }
  __device__ void finish(int) const {}
};

// ---- decay: one wavefront per candidate block ---------------------------------------------------------------------
// A wave takes kGather consecutive candidates at a time: lanes 0 .. kGather-1 fetch their list entries and the entries'
// block slots with one vector load each (the dependent chain list -> entry -> block of all of them in flight together,
// as in k_integrate), the wave walks them with v_readlane; the four 1 KiB chunks of a block are requested before the
// first is looked at.  (One candidate per wave-step with a scalar chain in front of every block: 3.8 TB/s read-only on
// the S-stress map; plain 4 KiB-per-wave streaming reaches 5.4.)
constexpr int kGather = 8;
#ifndef DSLAM_DECAY_WGS
#define DSLAM_DECAY_WGS 1024
#endif
constexpr int kDecayWgs = DSLAM_DECAY_WGS;
__global__ __launch_bounds__(256) void k_decay_blocks(const int *__restrict__ cand, const int *count_ptr,
                                                      const HashEntry *__restrict__ hash, uint4 *voxels16,
                                                      int max_weight, unsigned char *remove_flags,
                                                      unsigned char *remove_cand, int mark_empty) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256 + threadIdx.x) >> 6));
  const int n_waves = gridDim.x * 4;
  const int n = __builtin_amdgcn_readfirstlane(*count_ptr);
  for (int base = wave * kGather; base < n; base += n_waves * kGather) {
    int my_t = -1, my_ptr = -1;
    if (lane < kGather && base + lane < n) {
      my_t = cand[base + lane];
      my_ptr = hash[my_t].ptr;
    }
#pragma unroll 1
    for (int k = 0; k < kGather; k++) {
      const int ptr = __builtin_amdgcn_readlane(my_ptr, k);
      if (ptr < 0) continue;   // (past the end of the list)
      uint4 *blk = voxels16 + (size_t)ptr * (kBlock3 / 2);
      uint4 v[4];
#pragma unroll
      for (int j = 0; j < 4; j++) v[j] = load_nt(blk + j * 64 + lane);
      bool measured = false;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        bool ch = false;
        const unsigned w0 = (v[j].x >> 16) & 0xffu, w1 = (v[j].z >> 16) & 0xffu;
        if (w0 > 0 && (int)w0 <= max_weight) { v[j].x = kEmptyVoxelLo; v[j].y = kEmptyVoxelHi; ch = true; }
        if (w1 > 0 && (int)w1 <= max_weight) { v[j].z = kEmptyVoxelLo; v[j].w = kEmptyVoxelHi; ch = true; }
        measured |= (((v[j].x >> 16) & 0xffu) > 0) || (((v[j].z >> 16) & 0xffu) > 0);
        if (ch) store_nt(blk + j * 64 + lane, v[j]);
      }
      const bool any = __ballot(measured) != 0ull;
      if (!any && mark_empty && lane == 0) {
        remove_flags[__builtin_amdgcn_readlane(my_t, k)] = 1;
        remove_cand[base + k] = 1;
      }
    }
  }
}

// ---- removal list of a decay pass: ordered compaction over the CANDIDATE list -----------------------------------------
// cand[0..n) ascending in entry index; flag[i] = 1 for the candidates to release.  One launch: tiles of kSweepTile
// candidates taken by ticket, counts exchanged in-launch (dslam_bits.h look-back).  Tiles past the end publish zero and
// leave.  Consumes (clears) the flags.
__global__ __launch_bounds__(256) void k_compact_candidates(const int *__restrict__ cand, const int *count_ptr,
                                                            unsigned char *flag, int *__restrict__ out, int *total_out,
                                                            TileChain ch, SceneCounters *err_cnt) {
  __shared__ int red[8];
  __shared__ int s_ticket;
  const int n = __builtin_amdgcn_readfirstlane(*count_ptr);
  const int b = take_ticket(ch.ticket, ch.ticket_base, &s_ticket);   // (one tile per workgroup)
  if ((unsigned)b >= (unsigned)ch.n_tiles) return;   // (unsigned: see take_ticket)
  {
    const int i0 = b * kSweepTile + threadIdx.x * kSweepPer;
    unsigned m = 0;
    if (i0 < n) {
#pragma unroll
      for (int q = 0; q < kSweepPer / 4; q++) {
        if (i0 + q * 4 >= n) break;  // (the flag array is a multiple of 16 long)
        const unsigned w = *reinterpret_cast<const unsigned *>(flag + i0 + q * 4);
        if (w) {
          *reinterpret_cast<unsigned *>(flag + i0 + q * 4) = 0u;
          for (int k = 0; k < 4; k++)
            if ((w >> (8 * k)) & 0xffu) m |= 1u << (q * 4 + k);
        }
      }
    }
    int tot;
    int r = block_excl_scan<4>(__popc(m), red, tot);
    if (threadIdx.x == 0) publish1(ch.agg, b, ch.epoch, tot);
    const bool last = b == ch.n_tiles - 1;
    if (tot > 0 || last) {
      int offset;
      if (!lookback1(ch.agg, b, ch.epoch, red, offset) && threadIdx.x == 0) report_error(err_cnt, 2);
      if (last && threadIdx.x == 0) *total_out = offset + tot;
      r += offset;
      for (; m; m &= m - 1) out[r++] = cand[i0 + __ffs((int)m) - 1];
    }
  }
}

// ---- release: removal list -> pool; and the first removal of every affected bucket chain ------------------------------
// Two independent jobs in one launch: workgroups [0, kReleaseWgs) stream the released blocks (reset, slot back to the
// pool, rings cleared), the others decide for every removal whether it is the first released entry of its chain (the
// chain is still unmodified): those lead the rewrite in k_unlink.
constexpr int kReleaseWgs = 1024, kLeaderWgs = 64;

__global__ __launch_bounds__(256) void k_release_and_leaders(const int *__restrict__ rem, const SceneCounters *cnt,
                                                             const HashEntry *__restrict__ hash, uint4 *voxels16,
                                                             int *alloc_list, unsigned long long *masks, int *last_seen,
                                                             int words, int num_buckets, unsigned mask,
                                                             const unsigned char *__restrict__ remove_flags,
                                                             int *__restrict__ leader_bucket) {
  const int n = cnt->remove_count;
  if (n == 0) return;
  if (blockIdx.x < kReleaseWgs) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256 + threadIdx.x) >> 6));
    const int n_waves = kReleaseWgs * 4;
    const int base = cnt->last_free;
    const uint4 empty2 = make_uint4(kEmptyVoxelLo, kEmptyVoxelHi, kEmptyVoxelLo, kEmptyVoxelHi);
    // (kGather removals per wave-step: their list entries and block slots are fetched by lanes 0 .. kGather-1 together,
    // which also put the slots back on the free stack; the wave then resets the blocks one after the other)
    for (int r0 = wave * kGather; r0 < n; r0 += n_waves * kGather) {
      int my_ptr = -1;
      if (lane < kGather && r0 + lane < n) {
        my_ptr = hash[rem[r0 + lane]].ptr;
        alloc_list[base + 1 + r0 + lane] = my_ptr;
        last_seen[my_ptr] = -1;
      }
#pragma unroll 1
      for (int k = 0; k < kGather; k++) {
        const int ptr = __builtin_amdgcn_readlane(my_ptr, k);
        if (ptr < 0) continue;
        uint4 *blk = voxels16 + (size_t)ptr * (kBlock3 / 2);
#pragma unroll
        for (int j = 0; j < 4; j++) store_nt(blk + j * 64 + lane, empty2);
        if (lane < 2 * words) masks[(size_t)ptr * 2 * words + lane] = 0ull;
      }
    }
    return;
  }
  for (int r = (blockIdx.x - kReleaseWgs) * 256 + threadIdx.x; r < n; r += kLeaderWgs * 256) {
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

// leaders rewrite their bucket chain once (DESIGN.md "batch release"); consumes (clears) the removal flags
__global__ __launch_bounds__(256) void k_unlink(const int *__restrict__ leader_bucket, const SceneCounters *cnt,
                                                HashEntry *hash, int num_buckets, unsigned char *remove_flags,
                                                unsigned char *freed_flags, unsigned char *vis_type, unsigned *vis_bits,
                                                unsigned *alloc_bits, int *maint_flags, int list_is_foreign) {
  // list_is_foreign: the render state's visible list is not "the entries with a type" at the moment (FindVisibleBlocks
  // wrote it, or it was uploaded): any release then rebuilds it from the types, as the batch release is defined --
  // a released or moved entry may sit in that list without having a type (found by the fuzz test, seed 60045)
  const int n = cnt->remove_count;
  bool touched_visible = false;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
    const int head = leader_bucket[r];
    if (head < 0) continue;
    if (list_is_foreign) touched_visible = true;
    int c = head, prev = -1;
    while (c >= 0) {
      const HashEntry e = load_entry(hash, c);
      const int next = (e.offset >= 1) ? num_buckets + e.offset - 1 : -1;
      if (remove_flags[c]) {
        remove_flags[c] = 0;
        if (c != head) freed_flags[c - num_buckets] = 1;
        store_entry(hash, c, 0, 0, 0, 0, -2);
        bit_clear(alloc_bits, c);
        if (vis_type && vis_type[c] != 0) { touched_visible = true; vis_type[c] = 0; bit_clear(vis_bits, c); }
      } else {
        int cur = c;
        if (prev == -1) {
          if (c != head) {  // first survivor moves into the released bucket head
            store_entry(hash, head, e.pos[0], e.pos[1], e.pos[2], e.offset, e.ptr);
            if (e.ptr >= 0) bit_set(alloc_bits, head);
            bit_clear(alloc_bits, c);
            if (vis_type && vis_type[c] != 0) {  // (the released head's own type went to 0 a moment ago, by this lane)
              touched_visible = true;
              vis_type[head] = vis_type[c];
              bit_set(vis_bits, head);
              vis_type[c] = 0;
              bit_clear(vis_bits, c);
            }
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
  if (__ballot(touched_visible) && (threadIdx.x & 63) == 0) maint_flags[0] = 1;  // (a visible list has to be rebuilt)
}

// Freed excess slots back onto the excess free list in ascending slot order (single-pass ordered compaction: tile counts
// exchanged in-launch, tiles taken by ticket), flags consumed; the tile that ends the list folds the whole removal pass
// into the pool counters.
__device__ __forceinline__ void push_freed_job(unsigned char *freed_flags, int n_excess, int *excess_list, SceneCounters *cnt,
                                               int count_as_slid, const TileChain &ch, int *red, int &s_ticket) {
  // (read before this workgroup publishes anything; only the last tile changes them)
  const int removed = __builtin_amdgcn_readfirstlane(cnt->remove_count);
  const int base_ex = __builtin_amdgcn_readfirstlane(cnt->last_free_ex);
  const int b = take_ticket(ch.ticket, ch.ticket_base, &s_ticket);   // (one tile per workgroup)
  if (removed == 0 || (unsigned)b >= (unsigned)ch.n_tiles) return;  // (nothing was released: no slot came free and no flag is set)
  {
    const int i0 = b * kSweepTile + threadIdx.x * kSweepPer;
    unsigned m = 0;
    if (i0 < n_excess) {
#pragma unroll
      for (int q = 0; q < kSweepPer / 4; q++) {
        if (i0 + q * 4 >= n_excess) break;
        const unsigned w = *reinterpret_cast<const unsigned *>(freed_flags + i0 + q * 4);
        if (w) {
          *reinterpret_cast<unsigned *>(freed_flags + i0 + q * 4) = 0u;
          for (int k = 0; k < 4; k++)
            if (((w >> (8 * k)) & 0xffu) && i0 + q * 4 + k < n_excess) m |= 1u << (q * 4 + k);
        }
      }
    }
    int tot;
    int r = block_excl_scan<4>(__popc(m), red, tot);
    if (threadIdx.x == 0) publish1(ch.agg, b, ch.epoch, tot);
    const bool last = b == ch.n_tiles - 1;
    if (tot > 0 || last) {
      int offset;
      if (!lookback1(ch.agg, b, ch.epoch, red, offset) && threadIdx.x == 0) report_error(cnt, 2);
      r += offset + base_ex + 1;
      for (; m; m &= m - 1) excess_list[r++] = i0 + __ffs((int)m) - 1;
      if (last) {
        // every other tile's pushes target slots above the old top, and every other workgroup read the counters before it
        // published (this tile has just seen all their counts)
        __syncthreads();
        if (threadIdx.x == 0) {
          cnt->last_free += removed;
          cnt->last_free_ex = base_ex + offset + tot;
          if (count_as_slid) cnt->slid_blocks += removed; else cnt->decayed_blocks += removed;
          cnt->remove_count = 0;
          cnt->freed_excess = 0;
        }
      }
    }
  }
}

// rebuild of a render state's visible list from its types, only when the pass took an entry out of it: the entries with
// a bit in the render state's vis_bits, ascending.  Tile = 256 words, taken by ticket; counts exchanged in one look-back;
// the entries leave through an LDS list, one per lane and round (coalesced stores, full words of the excess area spread
// over the workgroup).
// `gen` = the generation bit of the render state's last allocation pass.  An entry that did not fit into the list of that
// pass (a visible list is capped at the pool size) kept its 1 / 2 with the NEXT pass' bit -- upstream leaves such a type in
// place without re-arming it, so it counts as marked again.  Once the rebuilt, shorter list has room for it, it is an
// ordinary listed entry, which upstream's next pass re-arms as 3: its byte gets the last pass' bit here (found by the
// fuzz test, seed 10744: list full after an allocation-only pass, then a window pop, then a fusion).
struct RebuildParams {
  const unsigned *vis_bits;   // null: no render state, nothing to rebuild
  unsigned char *vis_type;
  int *ids;
  int capacity;
  RenderCounters *rc;
  int *maint_flags;
  unsigned gen;
  int force;
  SceneCounters *cnt;         // (error flag) may be null
};

__device__ __forceinline__ void rebuild_visible_job(const RebuildParams &q, const TileChain &ch, int *red, int &s_ticket,
                                                    unsigned short *s_list) {
  // (every workgroup reads the flag before it takes its tile, hence before anything is published; the last tile re-arms it)
  const bool open = q.force || __builtin_amdgcn_readfirstlane(q.maint_flags[0]) != 0;
  const int b = take_ticket(ch.ticket, ch.ticket_base, &s_ticket);
  if (!open || (unsigned)b >= (unsigned)ch.n_tiles) return;
  const unsigned w = q.vis_bits[b * kCompactTileWords + threadIdx.x];
  int tot;
  const int rank = block_excl_scan<4>(__popc(w), red, tot);
  if (threadIdx.x == 0) publish1(ch.agg, b, ch.epoch, tot);
  const bool last = b == ch.n_tiles - 1;
  if (tot == 0 && !last) return;
  int before;
  if (!lookback1(ch.agg, b, ch.epoch, red, before) && threadIdx.x == 0 && q.cnt) report_error(q.cnt, 2);
  expand_bits(w, threadIdx.x * 32, rank, s_list);
  __syncthreads();
  const int t0 = b * (kCompactTileWords * 32);
  for (int j = threadIdx.x; j < tot; j += 256) {
    const int r = before + j;
    if (r >= q.capacity) break;
    const int t = t0 + s_list[j];
    q.ids[r] = t;
    const unsigned char ty = q.vis_type[t];
    if ((ty & 0x80u) != q.gen) q.vis_type[t] = (unsigned char)(q.gen | (ty & 0x7fu));
  }
  if (last && threadIdx.x == 0) {
    q.rc->no_visible = (before + tot) < q.capacity ? (before + tot) : q.capacity;
    q.maint_flags[0] = 0;
  }
}

// One launch, two independent jobs (each with its own ticket counter and its own channel of per-tile words): workgroups
// [0, push_wgs) put the freed excess slots back and fold the removal pass into the pool counters; the others rebuild the
// render state's visible list if the pass took an entry with a type.
__global__ __launch_bounds__(256) void k_push_freed_and_rebuild(unsigned char *freed_flags, int n_excess, int *excess_list,
                                                                SceneCounters *cnt, int count_as_slid, TileChain push_ch,
                                                                int push_wgs, RebuildParams q, TileChain vis_ch) {
  __shared__ int red[8];
  __shared__ int s_ticket;
  __shared__ unsigned short s_list[kCompactTileWords * 32];
  if ((int)blockIdx.x < push_wgs) push_freed_job(freed_flags, n_excess, excess_list, cnt, count_as_slid, push_ch, red, s_ticket);
  else rebuild_visible_job(q, vis_ch, red, s_ticket, s_list);
}

// scratch: candidate flags (table sized, overwritten whole by every selection), the lists
struct MaintScratch {
  unsigned char *rem_flags, *freed_flags, *rem_cand;
  int *cand_list, *rem_list, *leaders;
};
static MaintScratch carve(dslam_engine *e, int N) {
  MaintScratch m;
  m.rem_flags = e->rem_flags;
  m.freed_flags = e->freed_flags;
  m.rem_cand = e->rem_cand;
  m.cand_list = e->list_a;
  m.rem_list = e->list_b;
  m.leaders = e->list_d;
  (void)N;
  return m;
}

static RebuildParams rebuild_params(dslam_engine *e, dslam_render_state *r, bool force, SceneCounters *cnt) {
  RebuildParams q;
  q.vis_bits = r ? r->vis_bits : nullptr;
  q.vis_type = r ? r->visible_type : nullptr;
  q.ids = r ? r->visible_ids : nullptr;
  q.capacity = r ? r->n_local : 0;
  q.rc = r ? r->counters : nullptr;
  q.maint_flags = e->maint_flags;
  q.gen = r ? (unsigned)r->gen : 0u;
  q.force = force ? 1 : 0;
  q.cnt = cnt;
  return q;
}

// the swapping paths call this after they have changed types themselves: the rebuild alone, forced -- it also takes the
// "marked again" bit off entries that had not fitted into the last pass' list and are listed now
static int rebuild_visible_list(dslam_engine *e, dslam_render_state *r) {
  int grid;
  const TileChain vis_ch = next_chain(e, bit_tiles(r->n_entries) * (kBitTileWords / kCompactTileWords), &grid, true);
  TileChain push_ch = vis_ch;   // (no push job in this launch)
  push_ch.n_tiles = 0;
  hipLaunchKernelGGL(k_push_freed_and_rebuild, dim3(grid), dim3(256), 0, e->stream, (unsigned char *)nullptr, 0, (int *)nullptr,
                     (SceneCounters *)nullptr, 0, push_ch, 0, rebuild_params(e, r, true, nullptr), vis_ch);
  dbg_sync(e, "rebuild_visible_list");
  DSLAM_HIP(hipGetLastError());
  return DSLAM_OK;
}

// the tail of a removal pass: m.rem_list[0 .. remove_count) (ascending entry index, flags set in m.rem_flags) ->
// release + leaders, unlink, then ONE launch for the free-list pushes + counters and the visible-list rebuild (if a
// visible entry went)
static int release_listed(dslam_engine *e, dslam_scene *s, dslam_render_state *r, const MaintScratch &m, int count_as_slid) {
  hipLaunchKernelGGL(k_release_and_leaders, dim3(kReleaseWgs + kLeaderWgs), dim3(256), 0, e->stream, m.rem_list, s->counters,
                     s->hash, reinterpret_cast<uint4 *>(s->voxels), s->alloc_list, s->masks, s->last_seen, s->history_words,
                     s->p.num_buckets, (unsigned)(s->p.num_buckets - 1), m.rem_flags, m.leaders);
  dbg_sync(e, "k_release_and_leaders");
  hipLaunchKernelGGL(k_unlink, dim3(256), dim3(256), 0, e->stream, m.leaders, s->counters, s->hash, s->p.num_buckets,
                     m.rem_flags, m.freed_flags, r ? r->visible_type : (unsigned char *)nullptr,
                     r ? r->vis_bits : (unsigned *)nullptr, s->alloc_bits, e->maint_flags, (r && !r->types_follow_list) ? 1 : 0);
  dbg_sync(e, "k_unlink");
  {
    int push_grid, vis_grid = 0;
    const TileChain push_ch = next_chain(e, (s->p.num_excess + kSweepTile - 1) / kSweepTile, &push_grid);
    TileChain vis_ch = push_ch;
    vis_ch.n_tiles = 0;
    if (r) vis_ch = next_chain(e, bit_tiles(r->n_entries) * (kBitTileWords / kCompactTileWords), &vis_grid, true);
    hipLaunchKernelGGL(k_push_freed_and_rebuild, dim3(push_grid + vis_grid), dim3(256), 0, e->stream, m.freed_flags,
                       s->p.num_excess, s->excess_list, s->counters, count_as_slid, push_ch, push_grid,
                       rebuild_params(e, r, false, s->counters), vis_ch);
  }
  dbg_sync(e, "k_push_freed_and_rebuild");
  DSLAM_HIP(hipGetLastError());
  return DSLAM_OK;
}

// m.cand_list[0 .. swap_count) holds the selection: decay those blocks, release the ones left without a measured voxel
static int decay_candidates(dslam_engine *e, dslam_scene *s, dslam_render_state *r, const MaintScratch &m,
                            int max_weight) {
  hipLaunchKernelGGL(k_decay_blocks, dim3(kDecayWgs), dim3(256), 0, e->stream, m.cand_list, &s->counters->swap_count, s->hash,
                     reinterpret_cast<uint4 *>(s->voxels), max_weight, m.rem_flags, m.rem_cand, s->p.use_swapping ? 0 : 1);
  dbg_sync(e, "k_decay_blocks");
  DSLAM_HIP(hipGetLastError());
  if (s->p.use_swapping) return DSLAM_OK;  // entries of a swapping scene are never unlinked (ITMGlobalCache keys)
  {
    int grid;
    const TileChain ch = next_chain(e, (s->p.num_local_blocks + kSweepTile - 1) / kSweepTile, &grid);
    hipLaunchKernelGGL(k_compact_candidates, dim3(grid), dim3(256), 0, e->stream, m.cand_list, &s->counters->swap_count,
                       m.rem_cand, m.rem_list, &s->counters->remove_count, ch, s->counters);
  }
  dbg_sync(e, "k_compact_candidates");
  return release_listed(e, s, r, m, 0);
}

int launch_decay(dslam_engine *e, dslam_scene *s, dslam_render_state *r, int max_weight, int min_age, int force_all,
                 int q) {
  DSLAM_REQUIRE(!r || r->n_entries == s->n_entries, "render state was created for a different scene size");
  const int N = s->n_entries;
  int rc = ensure_scratch(e, N, s->p.num_local_blocks);
  if (rc) return rc;
  const MaintScratch m = carve(e, N);
  // swap_count doubles as the candidate count
  if (!force_all) {
    const int bits = 64 * s->history_words;
    const int newest = s->ring_next[q] - 1;
    int k = s->decay_cursor[q] > s->ring_head[q] ? s->decay_cursor[q] : s->ring_head[q];
    for (; k <= newest - min_age; k++) {
      SelDecayAged sel{s->hash, s->masks, s->history_words, q, k % bits};
      launch_bits_select(e, s->alloc_bits, N, sel, m.cand_list, s->p.num_local_blocks, &s->counters->swap_count, s->counters);
      if ((rc = decay_candidates(e, s, r, m, max_weight))) return rc;
    }
    if (k > s->decay_cursor[q]) s->decay_cursor[q] = k;
  } else {
    const int threshold = (s->frame_counter - 1) - min_age;
    SelDecaySweep sel{s->hash, s->last_seen, threshold};
    launch_bits_select(e, s->alloc_bits, N, sel, m.cand_list, s->p.num_local_blocks, &s->counters->swap_count, s->counters);
    dbg_sync(e, "select decay sweep");
    if ((rc = decay_candidates(e, s, r, m, max_weight))) return rc;
  }
  return DSLAM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// swapping (ITMSwappingEngine + ITMGlobalCache).  The global cache lives in page-locked host memory that the
// kernels read and write directly; the entry -> slot table and the slot counter live on the DEVICE, so the two swap
// steps of a ProcessFrame are kernels only (the host's part is to keep enough slabs mapped, ensure_slots).  The flush
// (SaveToGlobalMemory) and the window pop of a swapping scene still loop on host-read counts.
// ---------------------------------------------------------------------------------------------------------------
// SelSwapPending   swap state == 1 (needs the host copy merged): over the scene's swap1_bits   -- IntegrateGlobalIntoLocal
// SelSwapFresh     resident with state 0 (never visible since allocation): over alloc_bits      -- flush promotion
// SelSwapOut       state == 2, resident, not visible (or any visibility): over alloc_bits        -- SaveToGlobalMemory
struct SelSwapPending {
  DSLAM_SEL_NO_LOAD
  const unsigned char *swap_state;
  __device__ bool test(int t, const NoPayload &) const { return swap_state[t] == 1; }
  __device__ void prologue() const {}
  __device__ int emit(int, int, bool, const NoPayload &) const { return 0; }
  __device__ void finish(int) const {}
};
struct SelSwapFresh {
  DSLAM_SEL_NO_LOAD
  const HashEntry *hash;
  const unsigned char *swap_state;
  __device__ bool test(int t, const NoPayload &) const {
    const unsigned char st = swap_state[t];
    const int ptr = hash[t].ptr;   // (both requested before either is looked at)
    return (st == 0) & (ptr >= 0);
  }
  __device__ void prologue() const {}
  __device__ int emit(int, int, bool, const NoPayload &) const { return 0; }
  __device__ void finish(int) const {}
};
struct SelSwapOut {
  DSLAM_SEL_NO_LOAD
  const HashEntry *hash;
  const unsigned char *swap_state;
  const unsigned char *vis_type;   // null: whatever the visibility
  __device__ bool test(int t, const NoPayload &) const {
    const unsigned char st = swap_state[t];
    const int ptr = hash[t].ptr;
    const unsigned char ty = vis_type ? vis_type[t] : (unsigned char)0;
    return (st == 2) & (ptr >= 0) & (ty == 0);
  }
  __device__ void prologue() const {}
  __device__ int emit(int, int, bool, const NoPayload &) const { return 0; }
  __device__ void finish(int) const {}
};

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
// host slabs over PCIe (no slot: the entry has no host copy).  n = *n_dev (a count only the device knows) or n_host.
__global__ __launch_bounds__(256) void k_swap_merge(const int *__restrict__ ids, const int *__restrict__ slot_of_entry,
                                                    const int *__restrict__ n_dev, int n_host,
                                                    const HashEntry *__restrict__ hash, uint4 *voxels16,
                                                    uint4 *const *__restrict__ slabs, unsigned char *swap_state,
                                                    unsigned *swap1_bits, int only_unmerged, int maxW,
                                                    SceneCounters *stats) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256 + threadIdx.x) >> 6));
  const int n_waves = gridDim.x * 4;
  const int n = n_dev ? *n_dev : n_host;
  if (stats && blockIdx.x == 0 && threadIdx.x == 0) stats->swapped_in = n;
  for (int i = wave; i < n; i += n_waves) {
    const int t = ids[i];
    const unsigned char st = swap_state[t];
    if (only_unmerged && st == 2) continue;   // (a block leaving the window whose host copy is merged already)
    const int ptr = hash[t].ptr;
    const int slot = slot_of_entry[t];
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
    if (lane == 0) {
      swap_state[t] = 2;
      if (st == 1) bit_clear(swap1_bits, t);
    }
  }
}

// write selected blocks to their host slots (over PCIe, straight from the kernel), reset them, give their voxel-block
// slots back, mark the entries swapped out.  An entry that has no host slot yet takes the next one from the device-side
// counter (which slot an entry gets is not observable; that every entry keeps the one it got is what matters).
__global__ __launch_bounds__(256) void k_swap_pack(const int *__restrict__ ids, int *slot_of_entry,
                                                   const int *__restrict__ n_dev, int n_host, HashEntry *hash,
                                                   uint4 *voxels16, uint4 *const *__restrict__ slabs, int *alloc_list,
                                                   unsigned long long *masks, int *last_seen, int words,
                                                   unsigned char *swap_state, unsigned *swap1_bits, unsigned *alloc_bits,
                                                   unsigned char *vis_type, unsigned *vis_bits, SceneCounters *cnt) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256 + threadIdx.x) >> 6));
  const int n_waves = gridDim.x * 4;
  const int n = n_dev ? *n_dev : n_host;
  const int base = cnt->last_free;
  const uint4 empty2 = make_uint4(kEmptyVoxelLo, kEmptyVoxelHi, kEmptyVoxelLo, kEmptyVoxelHi);
  for (int i = wave; i < n; i += n_waves) {
    const int t = ids[i];
    const int ptr = hash[t].ptr;
    int slot = 0;
    if (lane == 0) {
      slot = slot_of_entry[t];
      if (slot < 0) {
        slot = atomicAdd(&cnt->next_slot, 1);
        slot_of_entry[t] = slot;
      }
    }
    slot = __builtin_amdgcn_readfirstlane(slot);
    uint4 *blk = voxels16 + (size_t)ptr * (kBlock3 / 2);
    uint4 *dst = stored_block(slabs, slot);
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
      bit_clear(alloc_bits, t);
      if (swap_state[t] == 1) bit_clear(swap1_bits, t);
      swap_state[t] = 0;
      if (vis_type && vis_type[t] != 0) { vis_type[t] = 0; bit_clear(vis_bits, t); }
    }
  }
}

__global__ void k_add_last_free(SceneCounters *cnt, const int *n_dev, int n_host, int add_slid, int set_stats) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int n = n_dev ? *n_dev : n_host;
    cnt->last_free += n;
    if (add_slid) cnt->slid_blocks += n;
    if (set_stats) cnt->swapped_out = n;
  }
}

__global__ void k_set_swap_stats(SceneCounters *cnt, int in, int out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { cnt->swapped_in = in; cnt->swapped_out = out; }
}

// select (ordered, capped at the transfer size): the list in m.cand_list, its length in counters->swap_count
// MODE 0 / 1 / 2 = SelSwapPending / SelSwapFresh / SelSwapOut
template <int MODE>
static int swap_select(dslam_engine *e, dslam_scene *s, const unsigned char *vis_type, const MaintScratch &m) {
  const int N = s->n_entries;
  int *count = &s->counters->swap_count;
  if (MODE == 0) launch_bits_select(e, s->swap1_bits, N, SelSwapPending{s->swap_state}, m.cand_list, kTransferBlocks, count, s->counters);
  else if (MODE == 1) launch_bits_select(e, s->alloc_bits, N, SelSwapFresh{s->hash, s->swap_state}, m.cand_list, kTransferBlocks, count, s->counters);
  else launch_bits_select(e, s->alloc_bits, N, SelSwapOut{s->hash, s->swap_state, vis_type}, m.cand_list, kTransferBlocks, count, s->counters);
  DSLAM_HIP(hipGetLastError());
  return DSLAM_OK;
}
// the same, and the length brought to the host (the flush loops until nothing is selected)
template <int MODE>
static int swap_select_to_host(dslam_engine *e, dslam_scene *s, const unsigned char *vis_type, const MaintScratch &m,
                               int *out_count) {
  int rc = swap_select<MODE>(e, s, vis_type, m);
  if (rc) return rc;
  int *host_count = reinterpret_cast<int *>(e->pinned) + 48;
  DSLAM_HIP(hipMemcpyAsync(host_count, &s->counters->swap_count, sizeof(int), hipMemcpyDeviceToHost, e->stream));
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

// Slabs for every slot the batch about to be queued may hand out (`need` at most).  The slot counter lives on the device;
// a copy of it is queued behind every packing batch, so whenever the stream has drained (always, between synchronous
// calls) the host knows it exactly; in between it counts with the upper bound and only waits when that bound runs
// into the end of the slabs.
static int ensure_slots(dslam_engine *e, dslam_scene *s, int need) {
  if (hipStreamQuery(e->stream) == hipSuccess) s->slot_bound = *s->next_slot_host;
  (void)hipGetLastError();  // (hipErrorNotReady is an answer, not a failure)
  while (s->slot_bound + need > (long long)s->slabs.size() * kSlabBlocks) {
    DSLAM_HIP(hipStreamSynchronize(e->stream));
    if (*s->next_slot_host < s->slot_bound) { s->slot_bound = *s->next_slot_host; continue; }
    int rc = add_slab(e, s);
    if (rc) return rc;
  }
  return DSLAM_OK;
}

// merge the host copies of the listed entries into the map; n_dev: length on the device, else n_host
static int merge_from_host(dslam_engine *e, dslam_scene *s, const int *ids, const int *n_dev, int n_host, bool set_stats,
                           bool only_unmerged = false) {
  hipLaunchKernelGGL(k_swap_merge, dim3(1024), dim3(256), 0, e->stream, ids, s->slot_dev, n_dev, n_host, s->hash,
                     reinterpret_cast<uint4 *>(s->voxels), s->slab_ptrs_dev, s->swap_state, s->swap1_bits, only_unmerged ? 1 : 0,
                     s->p.max_w, set_stats ? s->counters : nullptr);
  DSLAM_HIP(hipGetLastError());
  return DSLAM_OK;
}

// write the listed entries' blocks to the host store and release their voxel blocks (at most `upper` of them)
static int pack_to_host(dslam_engine *e, dslam_scene *s, dslam_render_state *r, const int *ids, const int *n_dev,
                        int n_host, int upper, int add_slid, bool set_stats) {
  int rc = ensure_slots(e, s, upper);
  if (rc) return rc;
  hipLaunchKernelGGL(k_swap_pack, dim3(1024), dim3(256), 0, e->stream, ids, s->slot_dev, n_dev, n_host, s->hash,
                     reinterpret_cast<uint4 *>(s->voxels), s->slab_ptrs_dev, s->alloc_list, s->masks,
                     s->last_seen, s->history_words, s->swap_state, s->swap1_bits, s->alloc_bits,
                     r ? r->visible_type : (unsigned char *)nullptr, r ? r->vis_bits : (unsigned *)nullptr, s->counters);
  hipLaunchKernelGGL(k_add_last_free, dim3(1), dim3(64), 0, e->stream, s->counters, n_dev, n_host, add_slid,
                     set_stats ? 1 : 0);
  DSLAM_HIP(hipGetLastError());
  DSLAM_HIP(hipMemcpyAsync(s->next_slot_host, &s->counters->next_slot, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  s->slot_bound += upper;
  return DSLAM_OK;  // (no wait: readers of the host store synchronise first)
}

// ProcessFrame's two swap steps run entirely on the device: selection, list, slots, transfer -- no host round trip
// (round 1 brought the ids to the host and the slots back, twice per keyframe)
int launch_swap_in(dslam_engine *e, dslam_scene *s, dslam_render_state *) {
  int rc = ensure_scratch(e, s->n_entries, s->p.num_local_blocks);
  if (rc) return rc;
  const MaintScratch m = carve(e, s->n_entries);
  if ((rc = swap_select<0>(e, s, nullptr, m))) return rc;
  return merge_from_host(e, s, m.cand_list, &s->counters->swap_count, 0, true);
}

int launch_swap_out(dslam_engine *e, dslam_scene *s, dslam_render_state *r, bool ignore_visibility) {
  int rc = ensure_scratch(e, s->n_entries, s->p.num_local_blocks);
  if (rc) return rc;
  const MaintScratch m = carve(e, s->n_entries);
  if ((rc = swap_select<2>(e, s, ignore_visibility ? nullptr : r->visible_type, m))) return rc;
  return pack_to_host(e, s, nullptr, m.cand_list, &s->counters->swap_count, 0, kTransferBlocks, 0, true);
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
    if ((rc = merge_from_host(e, s, m.cand_list, nullptr, n, false))) return rc;
  }
  while (true) {
    if ((rc = swap_select_to_host<1>(e, s, nullptr, m, &n))) return rc;
    if (n == 0) break;
    if ((rc = merge_from_host(e, s, m.cand_list, nullptr, n, false))) return rc;
  }
  int total = 0;
  while (true) {
    if ((rc = swap_select_to_host<2>(e, s, nullptr, m, &n))) return rc;
    if (n == 0) break;
    if ((rc = pack_to_host(e, s, nullptr, m.cand_list, nullptr, n, n, 0, false))) return rc;
    total += n;
  }
  hipLaunchKernelGGL(k_set_swap_stats, dim3(1), dim3(64), 0, e->stream, s->counters, 0, total);
  DSLAM_HIP(hipGetLastError());
  return DSLAM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// sliding window: pop the oldest list of ring q
// ---------------------------------------------------------------------------------------------------------------
int launch_slide_pop(dslam_engine *e, dslam_scene *s, dslam_render_state *r, int q) {
  DSLAM_REQUIRE(!r || r->n_entries == s->n_entries, "render state was created for a different scene size");
  const int N = s->n_entries;
  int rc = ensure_scratch(e, N, s->p.num_local_blocks);
  if (rc) return rc;
  const MaintScratch m = carve(e, N);
  const int bits = 64 * s->history_words;
  const int bit = (s->ring_head[q]++) % bits;
  if (s->decay_cursor[q] < s->ring_head[q]) s->decay_cursor[q] = s->ring_head[q];
  // the blocks whose rings are empty after clearing this list's bit leave the device, in ascending entry order
  if (!s->p.use_swapping) {
    // released: the list and the release pipeline's removal flags come out of one selection
    SelSlidePop sel{s->hash, s->masks, s->history_words, q, bit, m.rem_flags};
    launch_bits_select(e, s->alloc_bits, N, sel, m.rem_list, s->p.num_local_blocks, &s->counters->remove_count, s->counters);
    DSLAM_HIP(hipGetLastError());
    return release_listed(e, s, r, m, 1);
  }

  // scene with swapping: the blocks move to the host store, their entries stay (ptr = -1)
  SelSlidePop sel{s->hash, s->masks, s->history_words, q, bit, nullptr};
  launch_bits_select(e, s->alloc_bits, N, sel, m.cand_list, s->p.num_local_blocks, &s->counters->swap_count, s->counters);
  DSLAM_HIP(hipGetLastError());
  int *host_count = reinterpret_cast<int *>(e->pinned) + 48;
  DSLAM_HIP(hipMemcpyAsync(host_count, &s->counters->swap_count, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  DSLAM_HIP(hipStreamSynchronize(e->stream));   // (the host store must have slabs for every leaving block)
  const int total = *host_count;
  if (total == 0) return DSLAM_OK;
  // (1) leaving blocks whose host copy was never merged (state != 2): merge it first; (2) pack every leaving block.  The
  // kernels reach the page-locked store directly, so neither step needs upstream's transfer-sized batches.
  if ((rc = merge_from_host(e, s, m.cand_list, nullptr, total, false, true))) return rc;
  if ((rc = pack_to_host(e, s, r, m.cand_list, nullptr, total, total, 1, false))) return rc;
  if (r) return rebuild_visible_list(e, r);
  return DSLAM_OK;
}

}  // namespace dslam
