// dslam_bits.h -- ordered selection over a bitmap of the hash table (dslam_device.h, "bit-packed summaries"), in ONE launch.
//
// Every maintenance pass of the reference's engines is "for every hash entry in index order: if <condition> then ..."
// (decay candidates, blocks leaving the window, blocks to swap, live blocks to mesh, FindVisibleBlocks, the rebuild of a
// visible list).  On the device that is an ordered compaction; up to round 2 each one read all 1.18 M entries (19 MB, plus
// byte flags and a second launch for the ranks).  Here the candidates come from a bitmap -- allocated entries, entries
// with a type -- of which a set bit costs one sparse 16-byte read and a clear one nothing:
//   tile = 1024 words = 32768 entries, one 256-thread workgroup, 4 consecutive words per thread (thread order = entry order)
//   rank = popcounts + block scan + the counts of the tiles in front, exchanged inside the launch (one look-back; tiles are
//          taken by ticket, so a tile only waits for workgroups that are already running)
// A selection is a functor with
//   bool test(int t)                         is entry t (its bit is set in the source bitmap) selected?  May have side
//                                            effects on state that belongs to entry t alone.
//   void emit(int t, int rank, bool listed)  called for every selected entry in ascending order (listed: rank < capacity)
//   void finish(int total)                   called once (one thread of the tile that ends the table), behind every test
// `gate` (optional device flag): the selection runs only if *gate != 0 -- for passes that are needed only if an earlier
// kernel of the same call found work (the launch itself is unconditional: no host round trip).
#pragma once
#include "dslam_internal.h"

namespace dslam {

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

// out[0 .. min(total, capacity)) = the selected entries, ascending; *total_out = min(total, capacity) (written by the
// tile that ends the table).  `sel_bits` (optional): the selection as a bitmap as well.  *error_flags |= 2 if a count
// never arrived (cannot happen while the device makes progress; reported instead of spinning forever).
template <class Sel>
__global__ __launch_bounds__(256) void k_bits_select(const unsigned *__restrict__ src_bits, TileChain ch, Sel sel,
                                                     int *__restrict__ out, int capacity, int *total_out, unsigned *sel_bits,
                                                     int *error_flags, const int *gate) {
  __shared__ int red[4];
  __shared__ int s_ticket;
  const bool open = !gate || __builtin_amdgcn_readfirstlane(*gate) != 0;   // (every tile reads the flag before it publishes; finish() may re-arm it)
  // one tile per workgroup (the grid is the number of tiles): no loop around the barriers below
  const int b = take_ticket(ch.ticket, ch.ticket_base, &s_ticket);   // (taken even if the gate is shut: the host counts on it)
  if (!open || b >= ch.n_tiles) return;
  {
    const int w0 = b * kBitTileWords + threadIdx.x * 4;
    const uint4 src = *reinterpret_cast<const uint4 *>(src_bits + w0);
    uint4 pick = make_uint4(0, 0, 0, 0);
#pragma unroll 1
    for (int i = 0; i < 4; i++)
      for (unsigned m = sel4(src, i); m; m &= m - 1) {
        const int bit = __ffs((int)m) - 1;
        if (sel.test((w0 + i) * 32 + bit)) or4(pick, i, 1u << bit);
      }
    int tot;
    int r = block_excl_scan<4>(popc4(pick), red, tot);
    if (threadIdx.x == 0) publish1(ch.agg, b, ch.epoch, tot);
    if (sel_bits) *reinterpret_cast<uint4 *>(sel_bits + w0) = pick;
    const bool last = b == ch.n_tiles - 1;
    if (tot == 0 && !last) return;
    int before;
    if (!lookback1(ch.agg, b, ch.epoch, red, before) && threadIdx.x == 0 && error_flags) atomicOr(error_flags, 2);
    if (last && threadIdx.x == 0) {
      if (total_out) *total_out = (before + tot) < capacity ? (before + tot) : capacity;
      sel.finish(before + tot);
    }
    r += before;
#pragma unroll 1
    for (int i = 0; i < 4; i++)
      for (unsigned m = sel4(pick, i); m; m &= m - 1) {
        const int t = (w0 + i) * 32 + __ffs((int)m) - 1;
        if (r < capacity && out) out[r] = t;
        sel.emit(t, r, r < capacity);
        r++;
      }
  }
}

// host side: the chain of one launch (epoch + tickets); grid = one workgroup per tile, each takes exactly one ticket
inline TileChain next_chain(dslam_engine *e, int n_tiles, int *grid_out) {
  TileChain ch;
  if (++e->epoch == 0) e->epoch = 1;
  ch.agg = e->agg;
  ch.epoch = e->epoch;
  ch.ticket = e->ticket;
  ch.ticket_base = e->ticket_base;
  ch.n_tiles = n_tiles;
  const int grid = n_tiles > 0 ? n_tiles : 1;
  e->ticket_base += (unsigned)grid;
  *grid_out = grid;
  return ch;
}

template <class Sel>
inline void launch_bits_select(dslam_engine *e, const unsigned *src_bits, int n_entries, const Sel &sel, int *out, int capacity,
                               int *total_out, unsigned *sel_bits, int *error_flags, const int *gate = nullptr) {
  int grid;
  const TileChain ch = next_chain(e, bit_tiles(n_entries), &grid);
  hipLaunchKernelGGL(k_bits_select<Sel>, dim3(grid), dim3(256), 0, e->stream, src_bits, ch, sel, out, capacity, total_out,
                     sel_bits, error_flags, gate);
}

}  // namespace dslam
