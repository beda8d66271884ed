// alloc.hip -- scene reset, view conversion and AllocateSceneFromDepth for gfx950.
//
// Reference call sites: denseMapper->ResetScene (InfiniTamDriver.h:354-360), viewBuilder->UpdateView
// (InfiniTamDriver.cpp:280-288), denseMapper->ProcessFrame -> AllocateSceneFromDepth (InfiniTamDriver.h:187-192).
// Algorithm: SURVEY.md Appendix A.3, A.4, A.10.
//
// Allocation is bit-exact with the sequential CPU engine ("last writer in row-major pixel order wins", pool
// slots handed out in ascending hash-index order) although it runs wave-parallel, in two launches:
//   k_mark         per pixel, walk the +-mu segment in block units; misses do atomicMax(order_key[slot], pixel*cap+step+1)
//                  and set the slot's bit in a request bitmap; found entries get their type and a bit in `mark`
//   k_alloc_sweep  one pass over the BITMAPS (not the table): the final key of a requested slot names its winner, whose
//                  walk is replayed to the block it asked for; the r-th requesting entry in hash-index order gets
//                  voxelAllocationList[lastFree - r] (pool exhaustion follows the closed form derived in DESIGN.md);
//                  entries visible in the previous pass are re-tested against the frustum (by a job inside k_mark's
//                  launch); visibleEntryIDs is written ascending in hash index.  The ordered ranks are popcounts plus
//                  per-tile counts exchanged INSIDE the launch (see below).
// No host round trip between the phases: every count lives in device memory (SceneCounters/RenderCounters).
#include <cstdio>
#include <cstdlib>

#include "dslam_bits.h"

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
    if (i < n16) v[i] = empty2;   // (plain stores: the non-temporal policy is 1.5 % slower here, 198-201 -> 203-204 us per GiB)
  }
}

__global__ __launch_bounds__(256) void k_reset_tables(HashEntry *hash, int n_entries, int *alloc_list, int *last_seen,
                                                      int n_local, int *excess_list, int n_excess,
                                                      SceneCounters *cnt, int *err_host) {
  const int stride = gridDim.x * blockDim.x;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = tid; i < n_entries; i += stride) store_entry(hash, i, 0, 0, 0, 0, -2);
  for (int i = tid; i < n_local; i += stride) { alloc_list[i] = i; last_seen[i] = -1; }
  for (int i = tid; i < n_excess; i += stride) excess_list[i] = i;
  if (tid == 0) {
    SceneCounters c = {};
    c.err_host = err_host;
    c.last_free = n_local - 1;
    c.last_free_ex = n_excess - 1;
    *cnt = c;
  }
}

// test hook (dslam_debug_inject_device_error): a kernel that reports exactly what a failing pass would
__global__ void k_inject_error(SceneCounters *cnt, int bits) { report_error(cnt, bits); }
int launch_inject_error(dslam_engine *e, dslam_scene *s, int bits) {
  hipLaunchKernelGGL(k_inject_error, dim3(1), dim3(1), 0, e->stream, s->counters, bits);
  DSLAM_HIP(hipGetLastError());
  return DSLAM_OK;
}

int launch_scene_reset(dslam_engine *e, dslam_scene *s) {
  const size_t n16 = (size_t)s->p.num_local_blocks * kBlock3 / 2;
  const unsigned fill_wgs = (unsigned)((n16 + 256 * kFillUnroll - 1) / (256 * kFillUnroll));
  hipLaunchKernelGGL(k_fill_voxels, dim3(fill_wgs), dim3(256), 0, e->stream, reinterpret_cast<uint4 *>(s->voxels), n16);
  hipLaunchKernelGGL(k_reset_tables, dim3(2048), dim3(256), 0, e->stream, s->hash, s->n_entries, s->alloc_list,
                     s->last_seen, s->p.num_local_blocks, s->excess_list, s->p.num_excess, s->counters, e->err_host);
  DSLAM_HIP(hipMemsetAsync(s->masks, 0, (size_t)s->p.num_local_blocks * 2 * s->history_words * sizeof(unsigned long long),
                           e->stream));
  if (s->swap_state) {
    DSLAM_HIP(hipMemsetAsync(s->swap_state, 0, s->n_entries, e->stream));
    DSLAM_HIP(hipMemsetAsync(s->swap1_bits, 0, (size_t)bit_tiles(s->n_entries) * kBitTileWords * sizeof(unsigned), e->stream));
  }
  DSLAM_HIP(hipMemsetAsync(s->alloc_bits, 0, (size_t)bit_tiles(s->n_entries) * kBitTileWords * sizeof(unsigned), e->stream));
  DSLAM_HIP(hipGetLastError());
  return DSLAM_OK;
}

// alloc_bits from the table itself (after a table was uploaded: map restore, synthetic stress maps): the one pass that
// still reads every entry, off every hot path
__global__ __launch_bounds__(256) void k_build_alloc_bits(const HashEntry *__restrict__ hash, int n_entries, unsigned *alloc_bits) {
  const int w = blockIdx.x * 256 + threadIdx.x;   // (the grid covers whole bitmap tiles)
  unsigned out = 0;
  for (int k = 0; k < 32; k++) {
    const int t = w * 32 + k;
    if (t < n_entries && hash[t].ptr >= 0) out |= 1u << k;
  }
  alloc_bits[w] = out;
}

int launch_build_alloc_bits(dslam_engine *e, dslam_scene *s) {
  const int n_words = bit_tiles(s->n_entries) * kBitTileWords;
  hipLaunchKernelGGL(k_build_alloc_bits, dim3(n_words / 256), dim3(256), 0, e->stream, s->hash, s->n_entries, s->alloc_bits);
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
// TWO launches, and since round 3 neither reads the hash table as a whole: the 1.18 M-entry table is 19 MB, its keys and
// types another 6 MB, and a frame touches ~8 k entries of it.  What a pass needs to know about "all entries" lives in
// bitmaps of one bit per entry (147 KB each, dslam_device.h):
//   k_mark        per pixel: derive the float depth, walk the +-mu segment, mark found entries visible (type byte + a bit
//                 in `mark`), race for the order key of every missing block (atomicMax: the last pixel / step in
//                 row-major order wins) and set the slot's bit in q1 (empty bucket head: ordered request) or q2 (end of
//                 an occupied bucket's chain: excess request).  Its first workgroups run an independent job: the frustum
//                 re-test of every entry the render state held visible before the pass (`vis_bits`; set bits expanded
//                 into LDS and tested one per lane), outcome in `retest`.
//   k_alloc_sweep one pass over the bitmaps (144 tiles of 8192 entries; round 2: 288 tiles of 4096 TABLE entries): the
//                 r-th requesting entry in hash-index order gets voxelAllocationList[lastFree - r] -- ranks are
//                 popcounts; the winner of a slot is its final key, whose walk is replayed to the block it asked for;
//                 visible entries = retest | mark | committed requests, written ascending.  Per-tile counts travel in
//                 ONE in-launch look-back (both channels published before any work, tiles taken by ticket: a tile only
//                 waits for workgroups that are running).  Nothing else is exchanged between tiles: what a tile of the
//                 excess area needs to know about entries other tiles create there follows from the counts and the
//                 excess free list.
// What made the other launches disappear (round 2): no clearing pass (the sweep zeroes the keys it reads; allocType bytes
// of the previous pass are cleared through that pass' request bits), no second walk (replay), no re-arming pass (the
// visible types carry a generation bit, so "visible in the previous pass" is a non-zero byte with the other bit).
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
  unsigned char *alloc_type;   // allocType bytes of the previous pass are cleared through its request bits
  unsigned *q1, *q2, *mark;    // this pass' bitmaps (all zero when the pass starts)
  const unsigned *old_q1, *old_q2;  // the previous pass' request bits
  // the re-test job
  const unsigned *vis_bits;
  unsigned *retest;
  int retest_wgs, n_words;
  Mat4 M;
  float fx, fy, voxel_size;
  int swapping;
  unsigned long long *dbg;   // diagnostics (DSLAM_DBG_MARK=<file>): per workgroup 4 timestamps of its first wave (s_memtime)
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

// The re-test job of k_mark: workgroup = 64 words of the render state's visible bits (2048 entries).  An entry whose
// type byte carries the OTHER generation bit was visible in the previous pass (upstream re-arms it as 3 and tests it
// against the frustum); a byte with THIS pass' bit is visible whatever the test says (marked by a pixel of this launch, or
// a 1 / 2 that did not fit into the previous list and counts as marked again, see k_alloc_sweep).  The job only reads
// types: the bytes belong to the pixel lanes of this launch; the sweep writes the 3s.
// The set bits are expanded into an LDS list and tested one per lane and round: visible entries of the excess area sit
// in a few full words (excess slots are handed out contiguously), and a lane walking its own word would test 32 of them
// one after the other while its neighbours idle.
constexpr int kRetestWords = 64;   // words of vis_bits per workgroup of the re-test job: 2048 entries, 8 per thread
__device__ __forceinline__ void retest_job(const MarkParams &p) {
  __shared__ int red[4];
  __shared__ unsigned short s_list[kRetestWords * 32];
  __shared__ unsigned s_res[kRetestWords];
  const int w = blockIdx.x * kRetestWords + (threadIdx.x >> 2);   // (the job's workgroups cover the whole tiles of the bitmap)
  const int byte = threadIdx.x & 3;                               // this thread's 8 entries of the word
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // the pool tops as they are before this pass' commits (the sweep's last tile moves them)
    p.cnt->base_free = p.cnt->last_free;
    p.cnt->base_free_ex = p.cnt->last_free_ex;
  }
  for (unsigned m = ((p.old_q1[w] | p.old_q2[w]) >> (8 * byte)) & 0xffu; m; m &= m - 1)
    p.alloc_type[w * 32 + 8 * byte + __ffs((int)m) - 1] = 0;
  const unsigned bits = (p.vis_bits[w] >> (8 * byte)) & 0xffu;
  if (threadIdx.x < kRetestWords) s_res[threadIdx.x] = 0;
  int tot;
  const int rank = block_excl_scan<4>(__popc(bits), red, tot);
  expand_bits(bits, threadIdx.x * 8, rank, s_list);
  __syncthreads();
  const int t0 = blockIdx.x * (kRetestWords * 32);
  for (int j = threadIdx.x; j < tot; j += 256) {
    const int rel = s_list[j], t = t0 + rel;
    const unsigned char ty = p.vis_type[t];
    bool keep = false;
    if (ty != 0) {
      if ((ty & 0x80u) == p.gen) {
        keep = true;
      } else {
        const HashEntry e = load_entry(p.hash, t);
        bool vis, vis_enl;
        if (p.swapping) check_block_vis<true>(vis, vis_enl, e.pos[0], e.pos[1], e.pos[2], p.M, p.fx, p.fy, p.cx, p.cy, p.voxel_size, p.W, p.H);
        else check_block_vis<false>(vis, vis_enl, e.pos[0], e.pos[1], e.pos[2], p.M, p.fx, p.fy, p.cx, p.cy, p.voxel_size, p.W, p.H);
        keep = p.swapping ? vis_enl : vis;
      }
    }
    if (keep) atomicOr(&s_res[rel >> 5], 1u << (rel & 31));
  }
  __syncthreads();
  if (threadIdx.x < kRetestWords) p.retest[blockIdx.x * kRetestWords + threadIdx.x] = s_res[threadIdx.x];
}

// (diagnostics: [0] kernel entry, [1] exit, [2] depth pixel arrived, [3] walk over | wave_steps << 56.  The stamps are written
// from the kernel itself, not from a wrapper around a body function: with the argument struct handed on by reference the
// same code ran 23 us instead of 11.)
#define MARK_STAMP(i, extra) do { if (p.dbg && threadIdx.x == 0) p.dbg[(size_t)blockIdx.x * 4 + (i)] = __builtin_amdgcn_s_memtime() | (extra); } while (0)
__global__ __launch_bounds__(256) void k_mark(MarkParams p) {
  MARK_STAMP(0, 0ull);
  // (re-test workgroups first: with the pixel workgroups in front -- the launch ends with one of them -- the frame is 1.6 us
  // LONGER, 137.2 -> 138.9 us, four alternations)
  if ((int)blockIdx.x < p.retest_wgs) { retest_job(p); MARK_STAMP(1, 0ull); return; }
  const int idx = ((int)blockIdx.x - p.retest_wgs) * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
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
    report_error(p.cnt, 1);
    no_steps = p.step_cap;
  }
  int wave_steps = no_steps;
  for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(wave_steps, o, 64); wave_steps = v > wave_steps ? v : wave_steps; }
  MARK_STAMP(2, 0ull);

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
    bool need2 = false;   // ... and the slot is the end of an occupied bucket's chain (excess request)
    int h = 0;
    int found_h = -1;     // entry this lane's walk found at this step
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
        need2 = e.ptr >= -1;
      }
      if (found) found_h = h;
      pt.x += dir.x; pt.y += dir.y; pt.z += dir.z;
    }
    // A found entry that the render state did NOT hold visible before the pass gets a bit in `mark` (the sweep learns of
    // the others from vis_bits + their type bytes).  Only those: global atomics are dear on this part -- a bit for every
    // found entry made this kernel 48 us instead of 11 (137 us with a "bit already set?" load in front, which reads a line
    // the atomics keep taking away), and entries entering the view are a few hundred per frame.  vis_bits does not change
    // during this launch, so the test in front of the atomic is an ordinary cached load.  Neighbouring pixels find the
    // same entry: a lane whose left neighbour found it too leaves the bit to that lane.
    {
      const int left = __shfl_up(found_h, 1, 64);
      if (found_h >= 0 && !(lane > 0 && left == found_h)) {
        const unsigned bit = 1u << (found_h & 31);
        if (!(p.vis_bits[found_h >> 5] & bit)) atomicOr(&p.mark[found_h >> 5], bit);
      }
    }
    const unsigned key = (unsigned)idx * (unsigned)p.step_cap + (unsigned)i + 1u;
    // Neighbouring pixels ask for the same slot, and same-address atomics serialise at ~10 ns each on this part.  Keys
    // grow with the pixel index, so within a wavefront the highest lane of each group of equal slots holds the group's
    // maximum: only that lane issues the atomicMax (and the request bit).
    unsigned long long todo = __ballot(need);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int hl = __shfl(h, leader, 64);
      const unsigned long long grp = __ballot(need && h == hl);
      if (lane == 63 - __clzll((long long)grp)) {
        atomicMax(&p.keys[h], key);
        atomicOr(&(need2 ? p.q2 : p.q1)[h >> 5], 1u << (h & 31));
      }
      todo &= ~grp;
    }
  }
  MARK_STAMP(3, (unsigned long long)wave_steps << 56);
  MARK_STAMP(1, 0ull);
}
#undef MARK_STAMP

struct SweepParams {
  HashEntry *hash;
  int n_entries, num_buckets, n_tiles, n_words;   // n_words: whole tiles
  unsigned *keys;
  unsigned char *alloc_type;
  short4 *coords;
  const int *alloc_list;
  const int *excess_list;
  unsigned char *vis_type;
  unsigned char *swap_state;
  unsigned *swap1_bits;
  SceneCounters *cnt;
  RenderCounters *rc;
  int *vis_hint;    // page-locked copy of rc->no_visible for the host (dslam_render_state::vis_hint); may be null
  int hint_min;     // ... which only wants to know whether the count is >= this (dslam_engine::push_job_min)
  int hint_big;     // ... and what the host's copy answered when this pass was launched
  int *visible_ids;
  int capacity;
  const unsigned *q1, *q2, *mark, *retest;   // this pass
  unsigned *oq1, *oq2, *omark;               // the other set: zeroed here for the next pass
  unsigned *vis_bits, *alloc_bits;
  int *born;        // (re-integration batch) per voxel-block slot: which pass allocated it; null otherwise
  int born_stamp;
  unsigned long long *agg_req, *agg_vis;
  unsigned epoch;
  unsigned *ticket;
  unsigned ticket_base;
  unsigned gen;
  int do_commit;
  // replay of a winner's walk
  const float *depth;
  int W, H;
  Mat4 invM;
  float inv_fx, inv_fy, cx, cy, mu, one_over_block;
  int cap_shift;  // step_cap = 1 << cap_shift
  unsigned long long *dbg;  // diagnostics (DSLAM_DBG_SWEEP=<file>): per tile 8 timestamps (s_memtime)
};

// Pools that run out during the pass (rare): which requests get a block is the sequential rule over ALL requests in
// hash-index order -- a type-1 request at entry t succeeds iff vr(t) = c1(t) + min(c2(t), availEx) < availVBA, a type-2
// request iff also c2(t) < availEx (c1, c2: requests before t).  A tile behind the point where the voxel pool
// ran dry cannot tell from per-tile counts how many requests in front of it succeeded, so it walks the request bitmaps
// itself: thread i takes a contiguous range of words, a block scan gives it (c1, c2) at the start of its range.
//   s1q: successful type-1 requests at words < limit_word whose entry is not in (retest | mark)
//   f1s: FAILED type-1 requests at words < limit_word whose entry is in (retest | mark) -- a request that finds no block
//        takes the entry off the visible list even if a stale type had put it there (upstream writes the type 0)
//   s2 : successful type-2 requests (everywhere)
__device__ void walk_requests(const SweepParams &p, int limit_word, int avail_vba, int avail_ex, int *lds, int &s1q, int &f1s,
                              int &s2) {
  const int nw = p.n_words, per = (nw + 255) / 256;
  const int w_lo = threadIdx.x * per, w_hi = (w_lo + per) < nw ? (w_lo + per) : nw;
  int c1 = 0, c2 = 0;
  for (int w = w_lo; w < w_hi; w++) { c1 += __popc(p.q1[w]); c2 += __popc(p.q2[w]); }
  int tot;
  int r1 = block_excl_scan<4>(c1, lds, tot);
  int r2 = block_excl_scan<4>(c2, lds, tot);
  int n1q = 0, n1f = 0, n2 = 0;
  for (int w = w_lo; w < w_hi; w++) {
    const unsigned a = p.q1[w], b = p.q2[w];
    if (!(a | b)) continue;
    const unsigned seen = p.retest[w] | p.mark[w];
    for (unsigned m = a | b; m; m &= m - 1) {
      const int bit = __ffs((int)m) - 1;
      const int vr = r1 + (r2 < avail_ex ? r2 : avail_ex);
      if ((a >> bit) & 1u) {
        if (w < limit_word) {
          if (vr < avail_vba) n1q += !((seen >> bit) & 1u);
          else n1f += (seen >> bit) & 1u;
        }
        r1++;
      } else {
        if (r2 < avail_ex && vr < avail_vba) n2++;
        r2++;
      }
    }
  }
  int v[4] = {n1q, n2, n1f, 0};
  block_sum4(v, lds);
  s1q = v[0];
  s2 = v[1];
  f1s = v[2];
}

// A request's rank-independent half: replay the winning pixel's walk up to the winning step -- the block it asked for --
// and leave allocType / blockCoords as upstream's pass does (and the order keys clean for the next pass).
__device__ __forceinline__ short4 replay_request(const SweepParams &p, int t, bool is2) {
  const unsigned kz = p.keys[t] - 1u;
  p.keys[t] = 0;
  const int pix = (int)(kz >> p.cap_shift), step = (int)(kz & ((1u << p.cap_shift) - 1u));
  const int py = pix / p.W, px = pix - py * p.W;
  Vec3 pt, dir;
  ray_segment(p.depth[pix], px, py, p.invM, p.inv_fx, p.inv_fy, p.cx, p.cy, p.mu, p.one_over_block, pt, dir);
  for (int st = 0; st < step; st++) { pt.x += dir.x; pt.y += dir.y; pt.z += dir.z; }
  const short4 bc = make_short4((short)(int)floorf(pt.x), (short)(int)floorf(pt.y), (short)(int)floorf(pt.z), 1);
  p.alloc_type[t] = is2 ? 2 : 1;
  p.coords[t] = bc;
  return bc;
}

// ... and the half that needs its ranks (k1, k2: type-1 / type-2 requests in front of it in hash-index order)
template <bool SWAPPING>
__device__ __forceinline__ void commit_request(const SweepParams &p, int tile_first, int rel, bool is2, int k1, int k2, short4 bc, int base_free,
                                               int base_free_ex, int avail_vba, int avail_ex, unsigned *s_qv, unsigned *s_qf) {
  const int t = tile_first + rel;
  // voxel-block slots consumed by all earlier requests in hash-index order (closed form)
  const int vr = k1 + (k2 < avail_ex ? k2 : avail_ex);
  if (!is2) {
    const bool ok = p.do_commit && vr < avail_vba;
    if (ok) {
      const int slot = p.alloc_list[base_free - vr];
      store_entry(p.hash, t, bc.x, bc.y, bc.z, 0, slot);
      bit_set(p.alloc_bits, t);
      if (p.born) p.born[slot] = p.born_stamp;
    }
    // without the commit (onlyUpdateVisibleList) the request alone makes the entry "visible" this pass, like
    // upstream; with it, only if it got a block
    if (ok || !p.do_commit) {
      p.vis_type[t] = (unsigned char)(p.gen | 1u);
      atomicOr(&s_qv[rel >> 5], 1u << (rel & 31));
    } else {
      atomicOr(&s_qf[rel >> 5], 1u << (rel & 31));
    }
  } else if (p.do_commit && k2 < avail_ex && vr < avail_vba) {
    const int ex_off = p.excess_list[base_free_ex - k2];
    const int slot = p.alloc_list[base_free - vr];
    p.hash[t].offset = ex_off + 1;
    store_entry(p.hash, p.num_buckets + ex_off, bc.x, bc.y, bc.z, 0, slot);
    bit_set(p.alloc_bits, p.num_buckets + ex_off);
    if (p.born) p.born[slot] = p.born_stamp;
    // (its type byte and its place in the visible list are the business of the tile that owns the new entry)
  }
}

// One workgroup per tile of kSweepWords bitmap words, WPT consecutive words per thread.  The tile size is a trade: every
// phase below is a round trip (or a chain of them) that all tiles go through side by side, so at the bench's 8 k visible
// entries the launch takes as long as its slowest tile; the tile that ends the table holds the -- contiguous, mostly
// visible -- recent part of the excess area, and a map with 262 k visible entries has 7 k of them per 32768-entry tile.
// 8192-entry tiles: 144 for the default table (look-back: one word per lane), the excess area spread over 16 of them.
constexpr int kSweepWpt = 1;
constexpr int kSweepWords = 256 * kSweepWpt;

template <bool SWAPPING, int WPT>
__global__ __launch_bounds__(256) void k_alloc_sweep(SweepParams p) {
  constexpr int TW = 256 * WPT, TE = TW * 32;
  __shared__ int red[16];
  __shared__ int s_ticket;
  __shared__ unsigned s_newx[TW];   // entries other tiles' commits create in this tile (excess area)
  __shared__ unsigned s_mk[TW], s_qx[TW];   // the tile's mark bits / entries this pass made visible
  constexpr int kEmitWindow = TE < 8192 ? TE : 8192;
  __shared__ unsigned short s_list[kEmitWindow];
  constexpr int kReqWindow = 2048;
  __shared__ uint2 s_req[kReqWindow];       // requests of the tile: entry | type, ranks inside the tile
  __shared__ unsigned s_qv[TW], s_qf[TW];   // requests that made their entry visible / that found no block
  // (snapshot taken by k_mark)
  const int base_free = __builtin_amdgcn_readfirstlane(p.cnt->base_free), base_free_ex = __builtin_amdgcn_readfirstlane(p.cnt->base_free_ex);
  const int avail_vba = base_free + 1, avail_ex = base_free_ex + 1;
  const unsigned long long t_start = p.dbg ? __builtin_amdgcn_s_memtime() : 0ull;
  const int b = take_ticket(p.ticket, p.ticket_base, &s_ticket);   // (one tile per workgroup, in starting order)
  if ((unsigned)b >= (unsigned)p.n_tiles) return;   // (unsigned: see take_ticket)
#define STAMP(i) do { if (p.dbg && threadIdx.x == 0) p.dbg[(size_t)b * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
  if (p.dbg && threadIdx.x == 0) p.dbg[(size_t)b * 8] = t_start;
  STAMP(1);
  {
    const int lw0 = threadIdx.x * WPT;   // this thread's words inside the tile
    const int w0 = b * TW + lw0;
    unsigned q1[WPT], q2[WPT], mk[WPT], rt[WPT], pold[WPT];
#pragma unroll
    for (int i = 0; i < WPT; i++) { q1[i] = p.q1[w0 + i]; q2[i] = p.q2[w0 + i]; mk[i] = p.mark[w0 + i]; rt[i] = p.retest[w0 + i]; pold[i] = p.vis_bits[w0 + i]; }
    const int tile_first = b * TE;
    // `mark` only holds found entries that were not visible before (see k_mark); one that was is known by its type byte,
    // which carries this pass' generation bit.  The re-test job has accepted every such byte it saw -- but it ran while
    // the pixels were still marking, and an entry that fails the block frustum test can be marked all the same (a block
    // that cuts a corner of the image without one of its own corners inside).  So the bytes of the entries the job turned
    // down are looked at once more, now that the marking is over; dense, through the LDS list, because those entries
    // cluster like the visible ones do.  Usually there are none to a few dozen per tile.
    unsigned late[WPT];
    {
      int cpop = 0;
#pragma unroll
      for (int i = 0; i < WPT; i++) { late[i] = 0; s_qx[lw0 + i] = 0; cpop += __popc(pold[i] & ~rt[i]); }
      int ctot;
      const int crank = block_excl_scan<4>(cpop, red, ctot);
      if (ctot > 0) {
        for (int win = 0; win < ctot; win += kEmitWindow) {
          __syncthreads();
          int r = crank - win;
#pragma unroll
          for (int i = 0; i < WPT; i++)
            for (unsigned m = pold[i] & ~rt[i]; m; m &= m - 1) {
              if (r >= 0 && r < kEmitWindow) s_list[r] = (unsigned short)((lw0 + i) * 32 + __ffs((int)m) - 1);
              r++;
            }
          __syncthreads();
          const int n_win = (ctot - win) < kEmitWindow ? (ctot - win) : kEmitWindow;
          for (int j = threadIdx.x; j < n_win; j += 256) {
            const int rel = s_list[j];
            const unsigned char ty = p.vis_type[tile_first + rel];
            if (ty != 0 && (ty & 0x80u) == p.gen) atomicOr(&s_qx[rel >> 5], 1u << (rel & 31));
          }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < WPT; i++) late[i] = s_qx[lw0 + i];
        __syncthreads();   // (s_qx is used again further down)
      }
    }
    unsigned seen[WPT];
    int c1 = 0, c2 = 0, n_seen = 0, n_q1new = 0;
#pragma unroll
    for (int i = 0; i < WPT; i++) {
      seen[i] = rt[i] | mk[i] | late[i];
      c1 += __popc(q1[i]); c2 += __popc(q2[i]);
      n_seen += __popc(seen[i]); n_q1new += __popc(q1[i] & ~seen[i]);
    }
    STAMP(2);
    // ---- counts out first: nothing a tile publishes depends on another tile -------------------------------------------
    int r1, r2, tot1, tot2;
    block_excl_scan2<4>(c1, c2, red, r1, r2, tot1, tot2);
    int tv[4] = {n_seen, n_q1new, 0, 0};
    block_sum4(tv, red);
    if (threadIdx.x == 0) {
      publish(p.agg_req, b, p.epoch, tot1, tot2);
      publish(p.agg_vis, b, p.epoch, tv[0], tv[1]);
    }
    // the other set of bitmaps starts the next pass clean
#pragma unroll
    for (int i = 0; i < WPT; i++) { p.oq1[w0 + i] = 0; p.oq2[w0 + i] = 0; p.omark[w0 + i] = 0; s_newx[lw0 + i] = 0; }
    const bool last = b == p.n_tiles - 1;
    const bool has_excess = tile_first + TE > p.num_buckets || last;   // other tiles' commits may create entries here
    // requests of the tiles behind this one: only the tiles of the excess area need them (for the totals), and there
    // only excess requests exist -- counted straight from the bitmap
    int later2 = 0;
    if (has_excess)
      for (int w = (b + 1) * TW + threadIdx.x; w < p.n_words; w += 256) later2 += __popc(p.q2[w]);
    // The rank-independent half of a request -- its key, the winning pixel's depth, the walk to the block it asked for --
    // is worked out while the tile counts travel (when the tile has at most one request per lane: the usual case, a frame
    // allocates a few hundred blocks): two of the request's three dependent round trips are then behind the look-back's.
    const int req_tot = tot1 + tot2;
    auto expand_requests = [&](int win) {
      int k1 = r1, k2 = r2;   // ranks inside the tile
#pragma unroll
      for (int i = 0; i < WPT; i++) {
        const unsigned a1 = q1[i], a2 = q2[i];
        for (unsigned m = a1 | a2; m; m &= m - 1) {
          const int bit = __ffs((int)m) - 1;
          const bool is2 = (a2 >> bit) & 1u;
          const int idx = k1 + k2 - win;
          if (idx >= 0 && idx < kReqWindow)
            s_req[idx] = make_uint2((unsigned)((lw0 + i) * 32 + bit) | (is2 ? 0x80000000u : 0u), (unsigned)k1 | ((unsigned)k2 << 16));
          if (is2) k2++; else k1++;
        }
      }
    };
#pragma unroll
    for (int i = 0; i < WPT; i++) { s_qv[lw0 + i] = 0; s_qf[lw0 + i] = 0; }
    const bool staged = req_tot > 0 && req_tot <= 256;   // (uniform)
    bool st_has = false;
    uint2 st_rq = make_uint2(0, 0);
    short4 st_bc = make_short4(0, 0, 0, 0);
    if (staged) {
      expand_requests(0);
      __syncthreads();
      if ((int)threadIdx.x < req_tot) {
        st_has = true;
        st_rq = s_req[threadIdx.x];
        st_bc = replay_request(p, tile_first + (int)(st_rq.x & 0x7fffffffu), st_rq.x >> 31);
      }
    }
    STAMP(3);
    int pre[4];  // requests (type 1, type 2) and visible entries (retest | mark; new type-1 requests) in front of this tile
    if (!lookback2(p.agg_req, p.agg_vis, b, p.epoch, red, pre) && threadIdx.x == 0) report_error(p.cnt, 2);
    STAMP(4);
    int lv[4] = {later2, 0, 0, 0};
    if (has_excess) block_sum4(lv, red);
    const int all1 = pre[0] + tot1, all2 = pre[1] + tot2 + lv[0];   // (meaningful for has_excess tiles)
    // ---- does a pool run out?  (wave-uniform decisions) --------------------------------------------------------------
    const int vr_start = pre[0] + (pre[1] < avail_ex ? pre[1] : avail_ex);
    int vq_before = pre[3];   // visible type-1 requests in front of this tile that are not in (retest | mark)
    int seen_before = pre[2]; // entries in (retest | mark) in front of this tile that stay visible
    int succ2_all = 0;        // successful type-2 requests of the whole pass (has_excess tiles)
    if (p.do_commit) {
      const bool dry_before = vr_start > avail_vba;
      const bool dry_total = has_excess && all1 + (all2 < avail_ex ? all2 : avail_ex) > avail_vba;
      if (dry_before || dry_total) {
        int s1q, f1s, s2;
        walk_requests(p, b * TW, avail_vba, avail_ex, red, s1q, f1s, s2);
        if (dry_before) { vq_before = s1q; seen_before -= f1s; }
        succ2_all = s2;
      } else {
        succ2_all = all2 < avail_ex ? all2 : avail_ex;
      }
    }
    // ---- this tile's requests: replay, rank, commit -------------------------------------------------------------------
    // Dense: the requests are expanded into an LDS list with their ranks (a window of kReqWindow at a time) and taken one per
    // lane and round -- a request is a chain of dependent reads (key -> depth pixel -> free-list slot), and a lane that holds
    // two of them in its own words would walk it twice while the rest of the workgroup waits at the next barrier (per-tile
    // timeline: 4.6 + 1.8 us of a 16 us launch went there).
    if (!staged) {
      for (int win = 0; win < req_tot; win += kReqWindow) {
        __syncthreads();   // (s_qv / s_qf zeroed; the previous window read)
        expand_requests(win);
        __syncthreads();
        const int n_win = (req_tot - win) < kReqWindow ? (req_tot - win) : kReqWindow;
        for (int j = threadIdx.x; j < n_win; j += 256) {
          const uint2 rq = s_req[j];
          const int rel = (int)(rq.x & 0x7fffffffu);
          const bool is2 = rq.x >> 31;
          const short4 bc = replay_request(p, tile_first + rel, is2);
          commit_request<SWAPPING>(p, tile_first, rel, is2, pre[0] + (int)(rq.y & 0xffffu), pre[1] + (int)(rq.y >> 16), bc, base_free, base_free_ex,
                                   avail_vba, avail_ex, s_qv, s_qf);
        }
      }
    } else if (st_has) {
      commit_request<SWAPPING>(p, tile_first, (int)(st_rq.x & 0x7fffffffu), st_rq.x >> 31, pre[0] + (int)(st_rq.y & 0xffffu), pre[1] + (int)(st_rq.y >> 16),
                               st_bc, base_free, base_free_ex, avail_vba, avail_ex, s_qv, s_qf);
    }
    __syncthreads();
    unsigned qvis[WPT], qfail[WPT];   // type-1 requests that make their entry visible in this pass / that found no block
#pragma unroll
    for (int i = 0; i < WPT; i++) { qvis[i] = s_qv[lw0 + i]; qfail[i] = s_qf[lw0 + i]; }
    STAMP(5);
    // ---- entries other tiles create in the excess area: the first succ2_all slots off the excess free list ---------------
    int newx_before = 0;   // ... of them in front of this tile and not counted as (retest | mark) there
    if (has_excess) {
      for (int j = threadIdx.x; j < succ2_all; j += 256) {
        const int t = p.num_buckets + p.excess_list[base_free_ex - j];
        const int rel = t - tile_first;
        if (rel >= 0 && rel < TE) atomicOr(&s_newx[rel >> 5], 1u << (rel & 31));
        else if (rel < 0 && !(((p.retest[t >> 5] | p.mark[t >> 5]) >> (t & 31)) & 1u)) newx_before++;
      }
      int nv[4] = {newx_before, 0, 0, 0};
      block_sum4(nv, red);   // (has a barrier: s_newx is complete behind it)
      newx_before = nv[0];
    }
    STAMP(6);
    // ---- the visible list -----------------------------------------------------------------------------------------------
    unsigned vis[WPT];
    int vpop = 0;
#pragma unroll
    for (int i = 0; i < WPT; i++) {
      const unsigned newx = has_excess ? s_newx[lw0 + i] : 0u;
      vis[i] = (seen[i] & ~qfail[i]) | qvis[i] | newx;
      vpop += __popc(vis[i]);
      qvis[i] |= newx;   // (the pass made these visible itself: type 1)
    }
    int vis_tot;
    const int vis_rank = block_excl_scan<4>(vpop, red, vis_tot);   // (its barriers also order the request lanes' type bytes)
    const int vis_first = seen_before + vq_before + newx_before;
    // entries that are no longer visible (stores only: nothing to wait for)
#pragma unroll
    for (int i = 0; i < WPT; i++)
      for (unsigned m = pold[i] & ~vis[i]; m; m &= m - 1) p.vis_type[(w0 + i) * 32 + __ffs((int)m) - 1] = 0;
    // The visible entries are written densely: expanded into an LDS list (a window of kEmitWindow ranks at a time), then
    // one entry per lane and round -- coalesced list stores, and the visible entries of the excess area, which fill a few
    // words completely (excess slots are handed out contiguously), do not queue up behind a single lane.
    // The words are expanded by OTHER lanes than the ones that own them: word k by lane k % 256.  With several consecutive
    // words per lane the lanes that own the consecutive -- mostly visible -- recent entries of the excess area would each write
    // 128 list entries one after the other while the rest waits (32768-entry tiles: the tile that ends the table took 8.5 us
    // for its list where the others took 2.6).  (s_qv / s_qf are free again: the words and their ranks.)
    {
      int r = vis_rank;
#pragma unroll
      for (int i = 0; i < WPT; i++) {
        s_mk[lw0 + i] = mk[i]; s_qx[lw0 + i] = qvis[i];
        s_qv[lw0 + i] = vis[i]; s_qf[lw0 + i] = (unsigned)r;
        r += __popc(vis[i]);
      }
    }
    for (int win = 0; win < vis_tot; win += kEmitWindow) {
      __syncthreads();   // (s_mk / s_qx / words / ranks written; the previous window read)
#pragma unroll
      for (int q = 0; q < WPT; q++) {
        const int k = threadIdx.x + 256 * q;
        int r = (int)s_qf[k] - win;
        for (unsigned m = s_qv[k]; m; m &= m - 1) {
          if (r >= 0 && r < kEmitWindow) s_list[r] = (unsigned short)(k * 32 + __ffs((int)m) - 1);
          r++;
        }
      }
      __syncthreads();
      const int n_win = (vis_tot - win) < kEmitWindow ? (vis_tot - win) : kEmitWindow;
      // (four entries per lane at a time, their type bytes requested together: a tile of the excess area can hold a few
      // thousand visible entries -- several rounds where the others need one -- and a round is a round trip)
      constexpr int kEmitBatch = 4;
      for (int j0 = threadIdx.x; j0 < n_win; j0 += 256 * kEmitBatch) {
        int tt[kEmitBatch];
        unsigned fl[kEmitBatch];          // bit 0 in_mk, 1 in_qx, 2 in_x, 3 valid
        unsigned char old[kEmitBatch];
#pragma unroll
        for (int q = 0; q < kEmitBatch; q++) {
          const int j = j0 + q * 256;
          fl[q] = 0; tt[q] = 0; old[q] = 0;
          if (j < n_win) {
            const int rel = s_list[j];
            tt[q] = tile_first + rel;
            const bool in_mk = (s_mk[rel >> 5] >> (rel & 31)) & 1u, in_qx = (s_qx[rel >> 5] >> (rel & 31)) & 1u;
            const bool in_x = has_excess && ((s_newx[rel >> 5] >> (rel & 31)) & 1u);
            fl[q] = (in_mk ? 1u : 0u) | (in_qx ? 2u : 0u) | (in_x ? 4u : 0u) | 8u;
            if (!in_qx) old[q] = p.vis_type[tt[q]];   // (entries this pass made visible itself are type 1 whatever the byte says)
          }
        }
#pragma unroll
        for (int q = 0; q < kEmitBatch; q++) {
          if (!(fl[q] & 8u)) continue;
          const int t = tt[q], r = vis_first + win + j0 + q * 256;
          const bool in_mk = fl[q] & 1u, in_qx = fl[q] & 2u, in_x = fl[q] & 4u;
          if (r < p.capacity) {
            p.visible_ids[r] = t;
            if (in_x) {
              p.vis_type[t] = (unsigned char)(p.gen | 1u);   // (an entry another tile's commit created here)
            } else if (!in_mk && !in_qx) {
              // visible before, not marked now, inside the frustum: upstream's 3 (a byte with this pass' bit is a 1 / 2
              // that counts as marked again: it stays)
              if ((old[q] & 0x80u) != p.gen) p.vis_type[t] = (unsigned char)(p.gen | 3u);
            }
          } else {
            // no room in the list: upstream leaves the type in place without the entry being re-armed next pass, so
            // a 1 / 2 counts as marked again then (next pass' bit), a 3 is re-tested (this pass' bit)
            unsigned ty = 1;
            if (!in_qx) ty = (in_mk || (old[q] & 0x80u) == p.gen) ? (old[q] & 0x7fu) : 3u;
            p.vis_type[t] = (unsigned char)((ty == 3 ? p.gen : (p.gen ^ 0x80u)) | ty);
          }
          if (SWAPPING) {   // a visible entry's host copy (if it has one) is wanted back: IntegrateGlobalIntoLocal's state 1
            const unsigned char st = p.swap_state[t];
            if (st == 0) { p.swap_state[t] = 1; bit_set(p.swap1_bits, t); }
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < WPT; i++) p.vis_bits[w0 + i] = vis[i];
    STAMP(7);
    if (last && threadIdx.x == 0) {
      const int n = vis_first + vis_tot;
      p.rc->no_visible = n < p.capacity ? n : p.capacity;
      // (the host's copy is rewritten only when its answer would change: a store to host memory that the end of the launch
      // has to wait for costs every frame 0.6 us)
      if (p.vis_hint && (int)(n >= p.hint_min) != p.hint_big)
        __hip_atomic_store(p.vis_hint, n < p.capacity ? n : p.capacity, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (p.do_commit) {
        const int vr_all = all1 + (all2 < avail_ex ? all2 : avail_ex);
        const int succ_vba = vr_all < avail_vba ? vr_all : avail_vba;   // every success takes exactly one voxel-block slot
        p.cnt->last_free = base_free - succ_vba;
        p.cnt->base_free = base_free - succ_vba;   // (the pool top the re-allocation of a swapping scene counts from)
        p.cnt->last_free_ex = base_free_ex - succ2_all;
        p.cnt->alloc_failures = all1 + all2 - succ_vba;
      } else {
        p.cnt->alloc_failures = 0;
      }
    }
  }
}

#undef STAMP

// The visible list of a render state was replaced behind its types' back (FindVisibleBlocks into this render state, an
// uploaded list).  Upstream's next pass would leave every type as it is and set the LIST's entries to 3; in the
// generation encoding: a 1 / 2 that is to stay "marked" gets the coming pass' bit, a 3 keeps the old bit (re-tested),
// and the list's entries become old-bit 3s.
__global__ __launch_bounds__(256) void k_types_keep(unsigned char *vis_type, const unsigned *__restrict__ vis_bits, int n_words,
                                                    unsigned new_gen) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w >= n_words) return;
  for (unsigned m = vis_bits[w]; m; m &= m - 1) {
    const int t = w * 32 + __ffs((int)m) - 1;
    const unsigned ty = vis_type[t] & 0x7fu;
    if (ty) vis_type[t] = (unsigned char)((ty == 3 ? (new_gen ^ 0x80u) : new_gen) | ty);
  }
}
__global__ __launch_bounds__(256) void k_types_rearm(const int *__restrict__ ids, const RenderCounters *rc,
                                                     unsigned char *vis_type, unsigned *vis_bits, unsigned old_gen) {
  const int n = rc->no_visible;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int t = ids[i];
    vis_type[t] = (unsigned char)(old_gen | 3u);
    if (!((vis_bits[t >> 5] >> (t & 31)) & 1u)) bit_set(vis_bits, t);
  }
}

// reallocate swapped-out blocks that came back into view (useSwapping only): one pool, so the r-th request in
// hash-index order succeeds iff r < available.  Requests = visible entries (a bit in vis_bits) whose block is on the host:
// an ordered selection (dslam_bits.h) whose emit step hands the r-th of them voxelAllocationList[top - r].  The pool top
// the ranks refer to is the sweep's result, left in base_free by its last tile (nothing else moves it before finish()).
struct SelNeedsBlock {
  DSLAM_SEL_NO_LOAD
  HashEntry *hash;
  const int *alloc_list;
  unsigned *alloc_bits;
  SceneCounters *cnt;
  __device__ void prologue() const {}
  __device__ bool test(int t, const NoPayload &) const { return hash[t].ptr == -1; }
  __device__ int emit(int t, int rank, bool, const NoPayload &) const {
    const int base = cnt->base_free;
    if (rank <= base) {
      hash[t].ptr = alloc_list[base - rank];
      bit_set(alloc_bits, t);
    }
    return 0;
  }
  __device__ void finish(int total) const {
    const int base = cnt->base_free, avail = base + 1;
    cnt->last_free = base - (total < avail ? total : avail);
  }
};

static inline int ceil_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// words [lo, hi) of one of the alternating bitmap sets back to zero, and the allocType bytes its request bits stand for
// (only when scenes of different sizes share the engine: the pass that normally cleans a set covers its own table only)
__global__ __launch_bounds__(256) void k_clean_bits_tail(unsigned *q1, unsigned *q2, unsigned *mark, unsigned char *alloc_type,
                                                         int lo, int hi) {
  const int w = lo + blockIdx.x * 256 + threadIdx.x;
  if (w >= hi) return;
  for (unsigned m = q1[w] | q2[w]; m; m &= m - 1) alloc_type[w * 32 + __ffs((int)m) - 1] = 0;
  q1[w] = 0; q2[w] = 0; mark[w] = 0;
}

// steps along the +-mu segment: ceil(2 * |segment| in blocks) = ceil(mu / (2 * voxel_size)) for a rigid pose; the order
// key holds pixel * cap + step in 32 bits (also asked by dslam_reintegrate_batch before it changes anything)
int alloc_step_cap(const dslam_scene *s, int W, int H, int *cap_out) {
  const int step_bound = (int)ceilf(s->p.mu / (2.0f * s->p.voxel_size)) + 2;
  const int cap = ceil_pow2(step_bound + 1);
  if ((double)W * H * cap >= 4294967295.0) {
    set_last_error("image size x ray steps exceeds the 32-bit allocation order key");
    return DSLAM_ERR_UNSUPPORTED;
  }
  *cap_out = cap;
  return DSLAM_OK;
}

int launch_allocate(dslam_engine *e, dslam_scene *s, const dslam_view *v, dslam_render_state *r, const float *M_d,
                    const float *intr, int only_update_visible_list, int *list_out, void *count_out) {
  const int W = v->w_d, H = v->h_d, N = s->n_entries;
  DSLAM_REQUIRE(r->n_entries == N, "render state was created for a different scene size");
  DSLAM_REQUIRE((N & 15) == 0, "num_buckets + num_excess must be a multiple of 16");
  int rc = ensure_scratch(e, N, s->p.num_local_blocks);
  if (rc) return rc;

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
  mp.alloc_type = e->alloc_type;
  if ((rc = alloc_step_cap(s, W, H, &mp.step_cap))) return rc;
  int cap_shift = 0;
  while ((1 << cap_shift) < mp.step_cap) cap_shift++;

  // this pass' generation bit; what the previous pass left visible carries the other one
  const unsigned old_gen = r->gen;
  r->gen ^= 0x80u;
  mp.gen = r->gen;
  const int n_words = bit_tiles(N) * kBitTileWords, n_tiles = n_words / kSweepWords;
  if (!r->types_follow_list) {
    hipLaunchKernelGGL(k_types_keep, dim3(n_words / 256), dim3(256), 0, e->stream, r->visible_type, r->vis_bits, n_words, (unsigned)r->gen);
    hipLaunchKernelGGL(k_types_rearm, dim3(64), dim3(256), 0, e->stream, r->visible_ids, r->counters, r->visible_type, r->vis_bits, old_gen);
    r->types_follow_list = true;
  }

  // the alternating bitmap sets: this pass writes `cur` (clean since the pass before last) and cleans `oth`
  const int cur = (int)(e->alloc_pass & 1u), oth = cur ^ 1;
  e->alloc_pass++;
  if (e->bits_dirty[oth] > n_words) {  // the pass that dirtied `oth` ran on a larger table than this one
    const int lo = n_words, hi = e->bits_dirty[oth];
    hipLaunchKernelGGL(k_clean_bits_tail, dim3((hi - lo + 255) / 256), dim3(256), 0, e->stream, e->bits_q1[oth], e->bits_q2[oth],
                       e->bits_mark[oth], e->alloc_type, lo, hi);
  }
  e->bits_dirty[oth] = 0;
  e->bits_dirty[cur] = n_words;

  mp.q1 = e->bits_q1[cur]; mp.q2 = e->bits_q2[cur]; mp.mark = e->bits_mark[cur];
  mp.old_q1 = e->bits_q1[oth]; mp.old_q2 = e->bits_q2[oth];
  mp.vis_bits = r->vis_bits; mp.retest = e->bits_retest;
  mp.retest_wgs = n_words / kRetestWords; mp.n_words = n_words;
  memcpy(mp.M.m, M_d, sizeof(float) * 16);
  mp.fx = intr[0]; mp.fy = intr[1]; mp.voxel_size = s->p.voxel_size;
  mp.swapping = s->p.use_swapping ? 1 : 0;
  const int pix_blocks = (W * H + 255) / 256;
  mp.dbg = nullptr;
  static const char *dbg_mark_file = getenv("DSLAM_DBG_MARK");
  static int dbg_mark_calls = 0;
  unsigned long long *dbg_mark_host = nullptr;
  if (dbg_mark_file && ++dbg_mark_calls == 60) {
    DSLAM_HIP(hipHostMalloc((void **)&dbg_mark_host, (size_t)(mp.retest_wgs + pix_blocks) * 32, hipHostMallocDefault));
    memset(dbg_mark_host, 0, (size_t)(mp.retest_wgs + pix_blocks) * 32);
    mp.dbg = dbg_mark_host;
  }
  hipLaunchKernelGGL(k_mark, dim3(mp.retest_wgs + pix_blocks), dim3(256), 0, e->stream, mp);
  dbg_sync(e, "k_mark");
  if (dbg_mark_host) {
    DSLAM_HIP(hipStreamSynchronize(e->stream));
    if (FILE *f = fopen(dbg_mark_file, "wb")) { fwrite(dbg_mark_host, 32, mp.retest_wgs + pix_blocks, f); fclose(f); }
    (void)hipHostFree(dbg_mark_host);
  }
  v->depth_dirty = false;

  SweepParams sp;
  sp.hash = s->hash; sp.n_entries = N; sp.num_buckets = s->p.num_buckets; sp.n_tiles = n_tiles; sp.n_words = n_words;
  sp.keys = e->order_keys; sp.alloc_type = e->alloc_type; sp.coords = e->block_coords;
  sp.alloc_list = s->alloc_list; sp.excess_list = s->excess_list;
  sp.vis_type = r->visible_type; sp.swap_state = s->swap_state; sp.swap1_bits = s->swap1_bits;
  sp.cnt = s->counters; sp.capacity = r->n_local;
  sp.rc = count_out ? reinterpret_cast<RenderCounters *>(count_out) : r->counters;
  sp.vis_hint = count_out ? nullptr : r->vis_hint;
  sp.hint_min = e->push_job_min;
  sp.hint_big = r->vis_hint && __atomic_load_n(r->vis_hint, __ATOMIC_RELAXED) >= e->push_job_min;
  sp.visible_ids = list_out ? list_out : r->visible_ids;
  sp.q1 = mp.q1; sp.q2 = mp.q2; sp.mark = mp.mark; sp.retest = e->bits_retest;
  sp.oq1 = e->bits_q1[oth]; sp.oq2 = e->bits_q2[oth]; sp.omark = e->bits_mark[oth];
  sp.vis_bits = r->vis_bits; sp.alloc_bits = s->alloc_bits;
  sp.born = s->alloc_born; sp.born_stamp = s->alloc_born_stamp;
  sp.agg_req = e->agg; sp.agg_vis = e->agg + e->agg_tiles;
  if (++e->epoch == 0) e->epoch = 1;
  sp.epoch = e->epoch;
  const int grid = n_tiles;   // one tile per workgroup, one ticket each
  (void)tickets_ok(e);
  sp.ticket = e->ticket; sp.ticket_base = e->ticket_base;
  e->ticket_base += (unsigned)grid;
  sp.gen = r->gen;
  sp.do_commit = only_update_visible_list ? 0 : 1;
  sp.depth = v->depth; sp.W = W; sp.H = H;
  sp.invM = mp.invM; sp.inv_fx = mp.inv_fx; sp.inv_fy = mp.inv_fy; sp.cx = mp.cx; sp.cy = mp.cy;
  sp.mu = mp.mu; sp.one_over_block = mp.one_over_block; sp.cap_shift = cap_shift;
  sp.dbg = nullptr;
  static const char *dbg_file = getenv("DSLAM_DBG_SWEEP");
  static int dbg_calls = 0;
  unsigned long long *dbg_host = nullptr;
  if (dbg_file && ++dbg_calls == 60) {
    DSLAM_HIP(hipHostMalloc((void **)&dbg_host, (size_t)n_tiles * 64, hipHostMallocDefault));
    memset(dbg_host, 0, (size_t)n_tiles * 64);
    sp.dbg = dbg_host;
  }
  if (s->p.use_swapping) hipLaunchKernelGGL((k_alloc_sweep<true, kSweepWpt>), dim3(grid), dim3(256), 0, e->stream, sp);
  else hipLaunchKernelGGL((k_alloc_sweep<false, kSweepWpt>), dim3(grid), dim3(256), 0, e->stream, sp);
  dbg_sync(e, "k_alloc_sweep");
  if (dbg_host) {
    DSLAM_HIP(hipStreamSynchronize(e->stream));
    if (FILE *f = fopen(dbg_file, "wb")) { fwrite(dbg_host, 64, n_tiles, f); fclose(f); }
    (void)hipHostFree(dbg_host);
  }
  if (s->p.use_swapping) {
    SelNeedsBlock sel{s->hash, s->alloc_list, s->alloc_bits, s->counters};
    launch_bits_select(e, r->vis_bits, N, sel, (int *)nullptr, N, (int *)nullptr, s->counters);
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
