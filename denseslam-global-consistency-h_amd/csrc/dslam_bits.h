// dslam_bits.h -- ordered selection over a bitmap of the hash table (dslam_device.h, "bit-packed summaries").
//
// Every maintenance pass of the reference's engines is "for every hash entry in index order: if <condition> then ..."
// (decay candidates, blocks leaving the window, blocks to swap, live blocks to mesh, FindVisibleBlocks, the rebuild of a
// visible list).  On the device that is an ordered compaction; up to round 2 each one read all 1.18 M entries (19 MB, plus
// byte flags).  Here the candidates come from a bitmap -- allocated entries, entries with a type -- of which a set bit
// costs one sparse read and a clear one nothing.  One launch (k_bits_select, below): the set bits of a tile are expanded
// into an LDS list and tested densely, the tiles hand their counts to each other inside the launch, the selected entries
// are emitted in index order by the lanes that tested them.
// A selection is a functor with
//   void prologue()                          an independent grid-stride job (stores nobody in the launch waits for)
//   Payload load(int t)                      what test / emit need to read about entry t (a hash entry, or nothing): the
//                                            kernels request it for four entries per lane before they look at the first,
//                                            so that the rounds of a crowded tile -- the one that ends the table holds the
//                                            whole dense part of the excess area -- do not each pay their own round trip
//   bool test(int t, const Payload &)        is entry t (its bit is set in the source bitmap) selected?  May have side
//                                            effects on state that belongs to entry t alone.
//   Staged stage(int t, const Payload &)     what emit can work out about a selected entry without knowing its rank (runs
//                                            while the tile counts travel between the workgroups)
//   int  emit(int t, int rank, bool listed, const Staged &)
//                                            called for every selected entry (listed: rank < capacity; rank ascending with
//                                            t); the return values are summed per tile (tile_sum_out)
//   void finish(int total)                   called once (one thread of the tile that ends the table), behind every emit of
//                                            that tile
#pragma once
#include "dslam_internal.h"

namespace dslam {

struct NoPayload {};
// DSLAM_SEL_NO_STAGE: for selections whose emit has nothing to prepare (it gets the payload);
// DSLAM_SEL_NO_LOAD: for selections whose test / emit read what they need themselves (implies NO_STAGE)
#define DSLAM_SEL_NO_STAGE typedef Payload Staged; __device__ const Payload &stage(int, const Payload &p) const { return p; }
#define DSLAM_SEL_NO_LOAD typedef NoPayload Payload; __device__ NoPayload load(int) const { return NoPayload(); } DSLAM_SEL_NO_STAGE
constexpr int kSelBatch = 4;            // entries per lane whose loads are in flight together

constexpr int kCompactTileWords = 256;  // (tiles of the list rebuild in maintain.hip)
#ifndef DSLAM_SEL_THREADS
#define DSLAM_SEL_THREADS 1024
#endif
#ifndef DSLAM_SEL_BITS
#define DSLAM_SEL_BITS 8
#endif
constexpr int kSelThreads = DSLAM_SEL_THREADS;
constexpr int kSelBits = DSLAM_SEL_BITS;         // a selection tile: kSelBits bits of the bitmap per thread
constexpr int kSelTileWords = kSelThreads * kSelBits / 32;
static inline int select_tiles(int n_entries) { return bit_tiles(n_entries) * (kBitTileWords / kSelTileWords); }

// the set bits of `w` as entry indices relative to the tile, ascending, to list[rank ...]
__device__ __forceinline__ int expand_bits(unsigned w, int rel0, int rank, unsigned short *list) {
  for (; w; w &= w - 1) list[rank++] = (unsigned short)(rel0 + __ffs((int)w) - 1);
  return rank;
}

// ---- tiles taken by ticket + one in-launch look-back (list-tile compactions of the release pipeline) ---------------------
struct TileChain {
  unsigned long long *agg;
  unsigned epoch;
  unsigned *ticket;
  unsigned ticket_base;
  int n_tiles;
  unsigned long long *dbg;   // diagnostics (DSLAM_DBG_SELECT=<file>): 8 timestamps per tile
};

// single-channel look-back: a 32-bit count per tile ({epoch, count} in one word); bounded spin
template <int THREADS = 256>
static __device__ __forceinline__ bool lookback1(const unsigned long long *agg, int n, unsigned epoch, int *lds /* [THREADS / 64] */, int &sum) {
  int s = 0;
  bool ok = true;
  for (int j = threadIdx.x; j < n; j += THREADS) {
    unsigned long long w = __hip_atomic_load(&agg[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while ((unsigned)(w >> 32) != epoch) {
      if (++spins > kSpinLimit) { ok = false; break; }
      __builtin_amdgcn_s_sleep(4);
      w = __hip_atomic_load(&agg[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s += (int)(unsigned)w;
  }
  for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
  const int all_ok = __syncthreads_and(ok ? 1 : 0);
  int tot = 0;
#pragma unroll
  for (int w = 0; w < THREADS / 64; w++) tot += lds[w];
  sum = __builtin_amdgcn_readfirstlane(tot);
  __syncthreads();
  return __builtin_amdgcn_readfirstlane(all_ok) != 0;
}
static __device__ __forceinline__ void publish1(unsigned long long *agg, int tile, unsigned epoch, int count) {
  __hip_atomic_store(&agg[tile], ((unsigned long long)epoch << 32) | (unsigned)count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// host side: the chain of one launch (epoch + tickets); grid = one workgroup per tile, each takes exactly one ticket
// `second`: the chain runs next to another one in the same launch (own counter, own channel of per-tile words)
inline TileChain next_chain(dslam_engine *e, int n_tiles, int *grid_out, bool second = false) {
  TileChain ch;
  ch.dbg = nullptr;
  (void)tickets_ok(e);   // (after a HIP failure anywhere in the process: bases re-read from the device)
  if (++e->epoch == 0) e->epoch = 1;
  ch.agg = second ? e->agg + e->agg_tiles : e->agg;
  ch.epoch = e->epoch;
  ch.ticket = second ? e->ticket + 1 : e->ticket;
  unsigned &base = second ? e->ticket_base2 : e->ticket_base;
  ch.ticket_base = base;
  ch.n_tiles = n_tiles;
  const int grid = n_tiles > 0 ? n_tiles : 1;
  base += (unsigned)grid;
  *grid_out = grid;
  return ch;
}


// One launch: tile = kSelTileWords words = 8192 entries per 1024-thread workgroup, tiles taken by ticket.
//   1. the set bits of the tile are expanded into an LDS list (thread = one byte of a word) and tested DENSELY, one candidate
//      per lane and round: excess entries are handed out contiguously, so some bitmap words are full while most are nearly
//      empty -- a lane that walks "its" word bit by bit ends up with 32 dependent gathers where its neighbours have none
//      (measured: 100 us for FindVisibleBlocks that way).  1024 threads, so that the tile that holds the dense part of
//      the excess area needs two rounds of four loads per lane at most.
//   2. the verdicts (wave ballots, by list position) give the ranks inside the tile; the tile's count goes out, the counts of
//      the tiles in front come in (one look-back: tiles are taken in starting order, so every word waited for belongs to a
//      workgroup that is running; the spin is bounded all the same);
//      while the counts travel, every lane works out what its emits do not need the rank for (Sel::stage);
//   3. the selected entries are emitted, each by the lane that tested it (the payload of the first round is still in
//      registers).
// Up to the middle of round 3 this was two launches (test -> bitmap + counts | compact), 7.5 + 7.8 us for GetImage's
// FindVisibleBlocks where this one takes 12.8: merely fused, with the look-back in place of the boundary, it took 14.4; the
// staging is what the fusion made possible.  Other shapes measured: 512 threads x 4096 entries
// 15.2 us (twice the tickets), 1024 x 4096 20.7 us (two 1024-thread workgroups do not share a CU).
template <class Sel>
__global__ __launch_bounds__(kSelThreads) void k_bits_select(const unsigned *__restrict__ src_bits, Sel sel, int *__restrict__ out,
                                                             int capacity, int *total_out, int *tile_sum_out, TileChain ch,
                                                             SceneCounters *err_cnt) {
  constexpr int kWaves = kSelThreads / 64;
  constexpr int kTileEntries = kSelTileWords * 32;
  constexpr int kRounds = (kTileEntries + kSelThreads * kSelBatch - 1) / (kSelThreads * kSelBatch);
  __shared__ int red[kWaves];
  __shared__ int s_ticket;
  __shared__ unsigned short s_list[kTileEntries];
  __shared__ unsigned s_pick[kTileEntries / 32];   // verdicts by list position
  __shared__ int s_pref[kTileEntries / 32];
  const unsigned long long t_start = ch.dbg ? __builtin_amdgcn_s_memtime() : 0ull;
  const int b = take_ticket(ch.ticket, ch.ticket_base, &s_ticket);
  if ((unsigned)b >= (unsigned)ch.n_tiles) return;   // (unsigned: a ticket in front of the host's base must not index anything)
#define DSLAM_SEL_STAMP(i) do { if (ch.dbg && threadIdx.x == 0) ch.dbg[(size_t)b * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
  if (ch.dbg && threadIdx.x == 0) ch.dbg[(size_t)b * 8] = t_start;
  DSLAM_SEL_STAMP(1);
  const bool last = b == ch.n_tiles - 1;
  const int tid = threadIdx.x, lane = tid & 63;
  // thread = kSelBits entries: field (tid % kPerWord) of word (tid / kPerWord)
  constexpr int kPerWord = 32 / kSelBits;
  const unsigned byte = (src_bits[b * kSelTileWords + tid / kPerWord] >> ((tid % kPerWord) * kSelBits)) & ((1u << kSelBits) - 1u);
  int tot;
  const int rank0 = block_excl_scan<kWaves>(__popc(byte), red, tot);
  if (tot == 0) {   // (most tiles of a sparse bitmap)
    if (tid == 0) {
      publish1(ch.agg, b, ch.epoch, 0);
      if (tile_sum_out) tile_sum_out[b] = 0;
    }
    if (!last) { sel.prologue(); return; }
  }
  DSLAM_SEL_STAMP(2);
  if (tid < kTileEntries / 32) s_pick[tid] = 0;
  expand_bits(byte, tid * kSelBits, rank0, s_list);
  __syncthreads();
  DSLAM_SEL_STAMP(3);
  const int t0 = b * kTileEntries;
  // ---- test: four candidates per lane in flight; the first round's payloads stay in registers for the emit --------------
  typename Sel::Payload pl0[kSelBatch];
  int tt0[kSelBatch];
  unsigned pass0 = 0;
#pragma unroll
  for (int q = 0; q < kSelBatch; q++) {
    const int j = tid + q * kSelThreads;
    tt0[q] = j < tot ? t0 + (int)s_list[j] : -1;
    if (tt0[q] >= 0) pl0[q] = sel.load(tt0[q]);
  }
#pragma unroll
  for (int q = 0; q < kSelBatch; q++) {
    const bool ok = tt0[q] >= 0 && sel.test(tt0[q], pl0[q]);
    pass0 |= ok ? (1u << q) : 0u;
    const unsigned long long m = __ballot(ok);
    if (lane == 0 && m) { const int wi = (tid + q * kSelThreads) >> 5; s_pick[wi] = (unsigned)m; s_pick[wi + 1] = (unsigned)(m >> 32); }
  }
  for (int rd = 1; rd < kRounds; rd++) {
    if (rd * kSelThreads * kSelBatch >= tot) break;   // (uniform)
    typename Sel::Payload pl[kSelBatch];
    int tt[kSelBatch];
#pragma unroll
    for (int q = 0; q < kSelBatch; q++) {
      const int j = tid + (rd * kSelBatch + q) * kSelThreads;
      tt[q] = j < tot ? t0 + (int)s_list[j] : -1;
      if (tt[q] >= 0) pl[q] = sel.load(tt[q]);
    }
#pragma unroll
    for (int q = 0; q < kSelBatch; q++) {
      const bool ok = tt[q] >= 0 && sel.test(tt[q], pl[q]);
      const unsigned long long m = __ballot(ok);
      if (lane == 0 && m) { const int wi = (tid + (rd * kSelBatch + q) * kSelThreads) >> 5; s_pick[wi] = (unsigned)m; s_pick[wi + 1] = (unsigned)(m >> 32); }
    }
  }
  __syncthreads();
  DSLAM_SEL_STAMP(4);
  // ---- ranks inside the tile; count out, counts of the tiles in front in ---------------------------------------------------
  int picked;
  {
    const int c = tid < kTileEntries / 32 ? __popc(s_pick[tid]) : 0;
    const int pre = block_excl_scan<kWaves>(c, red, picked);
    if (tid < kTileEntries / 32) s_pref[tid] = pre;
  }
  if (tid == 0) publish1(ch.agg, b, ch.epoch, picked);
  __syncthreads();   // (s_pref)
  DSLAM_SEL_STAMP(5);
  // what an emit can work out without its rank is worked out while the counts travel
  typename Sel::Staged st0[kSelBatch];
  int rk0[kSelBatch];
#pragma unroll
  for (int q = 0; q < kSelBatch; q++) {
    const int j = tid + q * kSelThreads;
    rk0[q] = 0;
    if ((pass0 >> q) & 1u) {
      rk0[q] = s_pref[j >> 5] + __popc(s_pick[j >> 5] & ((1u << (j & 31)) - 1u));
      st0[q] = sel.stage(tt0[q], pl0[q]);
    }
  }
  int before;
  if (!lookback1<kSelThreads>(ch.agg, b, ch.epoch, red, before) && tid == 0 && err_cnt) report_error(err_cnt, 2);
  DSLAM_SEL_STAMP(6);
  // ---- emit -----------------------------------------------------------------------------------------------------------------
  int sum = 0;
#pragma unroll
  for (int q = 0; q < kSelBatch; q++) {
    if (!((pass0 >> q) & 1u)) continue;
    const int r = before + rk0[q];
    if (r < capacity && out) out[r] = tt0[q];
    sum += sel.emit(tt0[q], r, r < capacity, st0[q]);
  }
  for (int rd = 1; rd < kRounds; rd++) {
    if (rd * kSelThreads * kSelBatch >= tot) break;
#pragma unroll
    for (int q = 0; q < kSelBatch; q++) {
      const int j = tid + (rd * kSelBatch + q) * kSelThreads;
      if (j >= tot || !((s_pick[j >> 5] >> (j & 31)) & 1u)) continue;
      const int t = t0 + (int)s_list[j];
      const int r = before + s_pref[j >> 5] + __popc(s_pick[j >> 5] & ((1u << (j & 31)) - 1u));
      if (r < capacity && out) out[r] = t;
      sum += sel.emit(t, r, r < capacity, sel.stage(t, sel.load(t)));
    }
  }
  if (tile_sum_out && tot != 0) {
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d, 64);
    __syncthreads();
    if (lane == 0) red[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0) {
      int v = 0;
      for (int w = 0; w < kWaves; w++) v += red[w];
      tile_sum_out[b] = v;
    }
  }
  DSLAM_SEL_STAMP(7);
  sel.prologue();   // (the independent job: stores nobody in this launch waits for, so behind everything that is waited for)
  if (last) {
    __syncthreads();   // (every emit of this tile is behind us)
    if (tid == 0) {
      if (total_out) *total_out = (before + picked) < capacity ? (before + picked) : capacity;
      sel.finish(before + picked);
    }
  }
#undef DSLAM_SEL_STAMP
}

// out[0 .. min(total, capacity)) = the selected entries, ascending; *total_out = min(total, capacity).
// err_cnt: error bit 1 is reported there if a tile count never arrived (bounded spin; report_error, dslam_device.h)
template <class Sel>
inline void launch_bits_select(dslam_engine *e, const unsigned *src_bits, int n_entries, const Sel &sel, int *out, int capacity,
                               int *total_out, SceneCounters *err_cnt, int *tile_sum_out = nullptr) {
  const int n_words = bit_tiles(n_entries) * kBitTileWords;
  int grid;
  TileChain ch = next_chain(e, n_words / kSelTileWords, &grid);
  // diagnostics: per-tile timeline of the 60th selection that has per-tile sums (GetImage's FindVisibleBlocks)
  static const char *dbg_file = getenv("DSLAM_DBG_SELECT");
  static int dbg_calls = 0;
  unsigned long long *dbg_host = nullptr;
  if (dbg_file && tile_sum_out && ++dbg_calls == 60 && hipHostMalloc((void **)&dbg_host, (size_t)grid * 64, hipHostMallocDefault) == hipSuccess) {
    memset(dbg_host, 0, (size_t)grid * 64);
    ch.dbg = dbg_host;
  }
  hipLaunchKernelGGL(k_bits_select<Sel>, dim3(grid), dim3(kSelThreads), 0, e->stream, src_bits, sel, out, capacity, total_out,
                     tile_sum_out, ch, err_cnt);
  if (dbg_host) {
    (void)hipStreamSynchronize(e->stream);
    if (FILE *f = fopen(dbg_file, "wb")) { fwrite(dbg_host, 64, grid, f); fclose(f); }
    (void)hipHostFree(dbg_host);
  }
}

}  // namespace dslam
