// dslam_bits.h -- ordered selection over a bitmap of the hash table (dslam_device.h, "bit-packed summaries").
//
// Every maintenance pass of the reference's engines is "for every hash entry in index order: if <condition> then ..."
// (decay candidates, blocks leaving the window, blocks to swap, live blocks to mesh, FindVisibleBlocks, the rebuild of a
// visible list).  On the device that is an ordered compaction; up to round 2 each one read all 1.18 M entries (19 MB, plus
// byte flags).  Here the candidates come from a bitmap -- allocated entries, entries with a type -- of which a set bit
// costs one sparse read and a clear one nothing.  Two launches, both balanced:
//   k_bits_test     tile = 32 words = 1024 entries per 256-thread workgroup.  The set bits of the tile are expanded into an
//                   LDS list and tested DENSELY, one candidate per lane and round: excess entries are handed out
//                   contiguously, so some bitmap words are full while most are nearly empty -- a lane that walks "its" word
//                   bit by bit ends up with 32 dependent gathers where its neighbours have none (measured: 100 us for
//                   FindVisibleBlocks that way).  Out: the selection as a bitmap + the tile's count.
//   k_bits_compact  tile = 256 words = 8192 entries.  Rank = sum of the test tiles' counts in front + popcounts; the
//                   selected entries are again expanded into LDS and emitted densely (coalesced list stores).
// No look-back, no spin, no ticket: the only hand-off is the kernel boundary.
// A selection is a functor with
//   void prologue()                          an independent grid-stride job run by the compaction launch (optional work)
//   Payload load(int t)                      what test / emit need to read about entry t (a hash entry, or nothing): the
//                                            kernels request it for four entries per lane before they look at the first,
//                                            so that the rounds of a crowded tile -- the one that ends the table holds the
//                                            whole dense part of the excess area -- do not each pay their own round trip
//   bool test(int t, const Payload &)        is entry t (its bit is set in the source bitmap) selected?  May have side
//                                            effects on state that belongs to entry t alone.
//   int  emit(int t, int rank, bool listed, const Payload &)
//                                            called for every selected entry (listed: rank < capacity; rank ascending with
//                                            t); the return values are summed per compaction tile (tile_sum_out)
//   void finish(int total)                   called once (one thread of the tile that ends the table), behind every emit of
//                                            that tile
// `gate` (optional device flag): the selection runs only if *gate != 0 -- for passes that are needed only if an earlier
// kernel of the same call found work (the launches themselves are unconditional: no host round trip).  The test launch
// copies the flag to gate[1], which the compaction launch obeys, so that finish() may re-arm gate[0].
#pragma once
#include "dslam_internal.h"

namespace dslam {

struct NoPayload {};
// for selections whose test / emit read what they need themselves
#define DSLAM_SEL_NO_LOAD typedef NoPayload Payload; __device__ NoPayload load(int) const { return NoPayload(); }
constexpr int kSelBatch = 4;            // entries per lane whose loads are in flight together

constexpr int kTestTileWords = 32;      // k_bits_test: 1024 entries per workgroup
constexpr int kCompactTileWords = 256;  // k_bits_compact: 8192 entries per workgroup

// the set bits of `w` as entry indices relative to the tile, ascending, to list[rank ...]
__device__ __forceinline__ int expand_bits(unsigned w, int rel0, int rank, unsigned short *list) {
  for (; w; w &= w - 1) list[rank++] = (unsigned short)(rel0 + __ffs((int)w) - 1);
  return rank;
}

template <class Sel>
__global__ __launch_bounds__(256) void k_bits_test(const unsigned *__restrict__ src_bits, Sel sel, unsigned *__restrict__ pick_bits,
                                                   int *__restrict__ tile_counts, int *gate) {
  __shared__ int red[4];
  __shared__ unsigned short s_list[kTestTileWords * 32];
  __shared__ unsigned s_pick[kTestTileWords];
  if (gate) {
    const int g = __builtin_amdgcn_readfirstlane(gate[0]);
    if (blockIdx.x == 0 && threadIdx.x == 0) gate[1] = g;
    if (g == 0) return;
  }
  // thread = 4 entries: nibble (tid & 7) of word (tid >> 3)
  const int word = blockIdx.x * kTestTileWords + (threadIdx.x >> 3);
  const unsigned nib = (src_bits[word] >> ((threadIdx.x & 7) * 4)) & 0xfu;
  if (threadIdx.x < kTestTileWords) s_pick[threadIdx.x] = 0;
  int tot;
  const int rank = block_excl_scan<4>(__popc(nib), red, tot);   // (its barriers also cover s_pick)
  expand_bits(nib, threadIdx.x * 4, rank, s_list);
  __syncthreads();
  const int t0 = blockIdx.x * (kTestTileWords * 32);
  for (int j0 = threadIdx.x; j0 < tot; j0 += 256 * kSelBatch) {
    typename Sel::Payload pl[kSelBatch];
    int rel[kSelBatch];
#pragma unroll
    for (int q = 0; q < kSelBatch; q++) {
      const int j = j0 + q * 256;
      rel[q] = j < tot ? (int)s_list[j] : -1;
      if (rel[q] >= 0) pl[q] = sel.load(t0 + rel[q]);
    }
#pragma unroll
    for (int q = 0; q < kSelBatch; q++)
      if (rel[q] >= 0 && sel.test(t0 + rel[q], pl[q])) atomicOr(&s_pick[rel[q] >> 5], 1u << (rel[q] & 31));
  }
  __syncthreads();
  if (threadIdx.x < kTestTileWords) {
    const unsigned p = s_pick[threadIdx.x];
    pick_bits[blockIdx.x * kTestTileWords + threadIdx.x] = p;
    int c = __popc(p);
    for (int d = 16; d > 0; d >>= 1) c += __shfl_xor(c, d, 64);   // (kTestTileWords = 32 lanes of wave 0)
    if (threadIdx.x == 0) tile_counts[blockIdx.x] = c;
  }
}

template <class Sel>
__global__ __launch_bounds__(256) void k_bits_compact(const unsigned *__restrict__ pick_bits, const int *__restrict__ tile_counts,
                                                      Sel sel, int *__restrict__ out, int capacity, int *total_out,
                                                      int *tile_sum_out, const int *gate) {
  __shared__ int red[4];
  __shared__ unsigned short s_list[kCompactTileWords * 32];
  if (gate && __builtin_amdgcn_readfirstlane(gate[1]) == 0) return;
  sel.prologue();
  const unsigned w = pick_bits[blockIdx.x * kCompactTileWords + threadIdx.x];
  // the test tiles in front of this tile: (kCompactTileWords / kTestTileWords) per compaction tile
  const int before = block_sum_strided(tile_counts, blockIdx.x * (kCompactTileWords / kTestTileWords), 1, red);
  int tot;
  const int rank = block_excl_scan<4>(__popc(w), red, tot);
  const bool last = blockIdx.x == gridDim.x - 1;
  if (tot == 0 && !last) {
    if (tile_sum_out && threadIdx.x == 0) tile_sum_out[blockIdx.x] = 0;
    return;
  }
  expand_bits(w, threadIdx.x * 32, rank, s_list);
  __syncthreads();
  const int t0 = blockIdx.x * (kCompactTileWords * 32);
  int sum = 0;
  for (int j0 = threadIdx.x; j0 < tot; j0 += 256 * kSelBatch) {
    typename Sel::Payload pl[kSelBatch];
    int tt[kSelBatch];
#pragma unroll
    for (int q = 0; q < kSelBatch; q++) {
      const int j = j0 + q * 256;
      tt[q] = j < tot ? t0 + (int)s_list[j] : -1;
      if (tt[q] >= 0) pl[q] = sel.load(tt[q]);
    }
#pragma unroll
    for (int q = 0; q < kSelBatch; q++) {
      if (tt[q] < 0) continue;
      const int r = before + j0 + q * 256;
      if (r < capacity && out) out[r] = tt[q];
      sum += sel.emit(tt[q], r, r < capacity, pl[q]);
    }
  }
  if (tile_sum_out) {
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) tile_sum_out[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  }
  if (last) {
    __syncthreads();   // (every emit of this tile is behind us)
    if (threadIdx.x == 0) {
      if (total_out) *total_out = (before + tot) < capacity ? (before + tot) : capacity;
      sel.finish(before + tot);
    }
  }
}

// out[0 .. min(total, capacity)) = the selected entries, ascending; *total_out = min(total, capacity)
template <class Sel>
inline void launch_bits_select(dslam_engine *e, const unsigned *src_bits, int n_entries, const Sel &sel, int *out, int capacity,
                               int *total_out, int *tile_sum_out = nullptr, int *gate = nullptr) {
  const int n_words = bit_tiles(n_entries) * kBitTileWords;
  hipLaunchKernelGGL(k_bits_test<Sel>, dim3(n_words / kTestTileWords), dim3(256), 0, e->stream, src_bits, sel, e->bits_tmp,
                     e->tile_counts, gate);
  hipLaunchKernelGGL(k_bits_compact<Sel>, dim3(n_words / kCompactTileWords), dim3(256), 0, e->stream, e->bits_tmp, e->tile_counts,
                     sel, out, capacity, total_out, tile_sum_out, gate);
}

// ---- tiles taken by ticket + one in-launch look-back (list-tile compactions of the release pipeline) ---------------------
struct TileChain {
  unsigned long long *agg;
  unsigned epoch;
  unsigned *ticket;
  unsigned ticket_base;
  int n_tiles;
};

// single-channel look-back: a 32-bit count per tile ({epoch, count} in one word); bounded spin
static __device__ __forceinline__ bool lookback1(const unsigned long long *agg, int n, unsigned epoch, int *lds4, int &sum) {
  int s = 0;
  bool ok = true;
  for (int j = threadIdx.x; j < n; j += 256) {
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
  if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = s;
  const int all_ok = __syncthreads_and(ok ? 1 : 0);
  sum = __builtin_amdgcn_readfirstlane(lds4[0] + lds4[1] + lds4[2] + lds4[3]);
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

}  // namespace dslam
