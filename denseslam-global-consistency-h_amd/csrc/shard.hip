// shard.hip -- the exchange step of the sharded global re-integration (BASELINE configs[4]): which voxel
// blocks did the batch touch, and moving exactly those between the ranks.
//
// Reference: DenseSlam::OnlineCorrection (DenseSlam.cpp:298-432) de-integrates and re-integrates keyframes after a
// pose-graph correction; here that batch runs on several GPUs, every rank updating the voxel blocks whose slot chunk
// (slot / chunk_blocks) % num_shards is its own.  Allocation is replicated and bit-identical on all ranks, so every rank
// knows every block the batch visited (the integration kernel marks a per-slot byte for each visible resident block
// BEFORE its shard test).  Hence no ids travel and no counts are exchanged: every rank derives the same per-shard
// lists of dirty slots, packs its own shard's blocks in list order, one all-gather (padded to the longest list) moves
// them, and the blocks of the other shards are put in place from the same lists.  Only blocks the batch touched move --
// not the used range of the pool, whose extent says nothing about where live blocks sit once decay, the sliding window or
// swapping have returned slots to the free list in arbitrary order.
#include "dslam_internal.h"

namespace dslam {

// "virtual" slot order: shard by shard, inside a shard ascending in slot.  N % (C * W) == 0.
struct ShardLayout {
  int n_local, shards, chunk, per_shard;  // per_shard = n_local / shards
};
__device__ __forceinline__ int virtual_to_slot(const ShardLayout &L, int v) {
  const int r = v / L.per_shard, rem = v - r * L.per_shard;
  const int j = rem / L.chunk;
  return (j * L.shards + r) * L.chunk + (rem - j * L.chunk);
}

// flags in virtual order, so that ONE ordered compaction yields every shard's ascending list, back to back
__global__ __launch_bounds__(256) void k_dirty_permute(const unsigned char *__restrict__ dirty, unsigned char *__restrict__ out,
                                                       ShardLayout L) {
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v < L.n_local) out[v] = dirty[virtual_to_slot(L, v)];
}

// per shard: number of dirty slots (one workgroup per shard)
__global__ __launch_bounds__(256) void k_dirty_counts(const unsigned char *__restrict__ vflags, ShardLayout L, int *__restrict__ counts) {
  __shared__ int red[4];
  const int r = blockIdx.x;
  int c = 0;
  for (int i = threadIdx.x; i < L.per_shard; i += 256) c += vflags[r * L.per_shard + i] != 0;
  for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[r] = red[0] + red[1] + red[2] + red[3];
}

// one wavefront per block: voxel blocks of `shard`, in list order, into a packed buffer (or back, for every OTHER shard)
__global__ __launch_bounds__(256) void k_dirty_pack(const int *__restrict__ vlist, const int *__restrict__ counts, int shard,
                                                    ShardLayout L, const uint4 *__restrict__ voxels16, uint4 *__restrict__ send,
                                                    int capacity) {
  const int lane = threadIdx.x & 63;
  const int wave = (int)((blockIdx.x * 256 + threadIdx.x) >> 6), n_waves = gridDim.x * 4;
  int off = 0;
  for (int r = 0; r < shard; r++) off += counts[r];
  const int n = counts[shard] < capacity ? counts[shard] : capacity;
  for (int i = wave; i < n; i += n_waves) {
    const uint4 *blk = voxels16 + (size_t)virtual_to_slot(L, vlist[off + i]) * (kBlock3 / 2);
    uint4 *dst = send + (size_t)i * (kBlock3 / 2);
#pragma unroll
    for (int j = 0; j < 4; j++) dst[j * 64 + lane] = blk[j * 64 + lane];
  }
}

__global__ __launch_bounds__(256) void k_dirty_unpack(const int *__restrict__ vlist, const int *__restrict__ counts, int skip_shard,
                                                      ShardLayout L, uint4 *__restrict__ voxels16, const uint4 *__restrict__ recv,
                                                      int stride_blocks) {
  const int lane = threadIdx.x & 63;
  const int wave = (int)((blockIdx.x * 256 + threadIdx.x) >> 6), n_waves = gridDim.x * 4;
  int off = 0;
  for (int r = 0; r < L.shards; r++) {
    const int n = counts[r] < stride_blocks ? counts[r] : stride_blocks;
    if (r != skip_shard)
      for (int i = wave; i < n; i += n_waves) {
        uint4 *blk = voxels16 + (size_t)virtual_to_slot(L, vlist[off + i]) * (kBlock3 / 2);
        const uint4 *src = recv + ((size_t)r * stride_blocks + i) * (kBlock3 / 2);
#pragma unroll
        for (int j = 0; j < 4; j++) blk[j * 64 + lane] = src[j * 64 + lane];
      }
    off += counts[r];
  }
}

int launch_dirty_plan(dslam_engine *e, dslam_scene *s, int num_shards, int chunk_blocks, int *counts_host) {
  const int N = s->p.num_local_blocks;
  DSLAM_REQUIRE(s->dirty, "dslam_scene_track_dirty has not been enabled on this scene");
  DSLAM_REQUIRE(num_shards >= 1 && num_shards <= 64 && chunk_blocks >= 1 && N % (num_shards * chunk_blocks) == 0,
                "num_local_blocks must be a multiple of num_shards * chunk_blocks (and num_shards <= 64)");
  int rc = ensure_scratch(e, s->n_entries, N);
  if (rc) return rc;
  const ShardLayout L = {N, num_shards, chunk_blocks, N / num_shards};
  unsigned char *vflags = reinterpret_cast<unsigned char *>(e->list_c);  // >= N bytes
  const int n_tiles = num_tiles(N);
  hipLaunchKernelGGL(k_dirty_permute, dim3((N + 255) / 256), dim3(256), 0, e->stream, s->dirty, vflags, L);
  hipLaunchKernelGGL(k_dirty_counts, dim3(num_shards), dim3(256), 0, e->stream, vflags, L, s->dirty_counts);
  // ordered compaction of the virtual flags: count per tile, then place (k_compact_apply_fused sums the preceding tiles)
  hipLaunchKernelGGL(k_flag_count, dim3(n_tiles), dim3(256), 0, e->stream, vflags, N, e->tile_counts);
  hipLaunchKernelGGL(k_compact_apply_fused, dim3(n_tiles), dim3(256), 0, e->stream, vflags, N, e->tile_counts, s->dirty_list, N,
                     s->dirty_counts + 64);
  DSLAM_HIP(hipGetLastError());
  int *host = reinterpret_cast<int *>(e->pinned) + 256;
  DSLAM_HIP(hipMemcpyAsync(host, s->dirty_counts, (size_t)num_shards * sizeof(int), hipMemcpyDeviceToHost, e->stream));
  DSLAM_HIP(hipStreamSynchronize(e->stream));
  for (int r = 0; r < num_shards; r++) counts_host[r] = host[r];
  s->dirty_shards = num_shards;
  s->dirty_chunk = chunk_blocks;
  return DSLAM_OK;
}

int launch_dirty_pack(dslam_engine *e, const dslam_scene *s, int shard, void *send_dev, int capacity_blocks) {
  DSLAM_REQUIRE(s->dirty && s->dirty_shards > 0, "dslam_shard_dirty_plan has not run on this scene");
  DSLAM_REQUIRE(shard >= 0 && shard < s->dirty_shards && send_dev && capacity_blocks >= 0, "bad argument");
  const ShardLayout L = {s->p.num_local_blocks, s->dirty_shards, s->dirty_chunk, s->p.num_local_blocks / s->dirty_shards};
  hipLaunchKernelGGL(k_dirty_pack, dim3(1024), dim3(256), 0, e->stream, s->dirty_list, s->dirty_counts, shard, L,
                     reinterpret_cast<const uint4 *>(s->voxels), reinterpret_cast<uint4 *>(send_dev), capacity_blocks);
  DSLAM_HIP(hipGetLastError());
  return DSLAM_OK;
}

int launch_dirty_unpack(dslam_engine *e, dslam_scene *s, int skip_shard, const void *recv_dev, int stride_blocks) {
  DSLAM_REQUIRE(s->dirty && s->dirty_shards > 0, "dslam_shard_dirty_plan has not run on this scene");
  DSLAM_REQUIRE(recv_dev && stride_blocks >= 0, "bad argument");
  const ShardLayout L = {s->p.num_local_blocks, s->dirty_shards, s->dirty_chunk, s->p.num_local_blocks / s->dirty_shards};
  hipLaunchKernelGGL(k_dirty_unpack, dim3(1024), dim3(256), 0, e->stream, s->dirty_list, s->dirty_counts, skip_shard, L,
                     reinterpret_cast<uint4 *>(s->voxels), reinterpret_cast<const uint4 *>(recv_dev), stride_blocks);
  DSLAM_HIP(hipGetLastError());
  return DSLAM_OK;
}

}  // namespace dslam
