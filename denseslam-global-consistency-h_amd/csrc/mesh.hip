// mesh.hip -- ITMMeshingEngine::MeshScene (SaveCurrSceneToMesh, reference DenseSlam.cpp:638-643): marching cubes over
// every allocated voxel block, SURVEY.md 8f N4.
//
// Upstream's CPU engine walks the hash table in entry order, each block in z, y, x order, and appends each cube's
// triangles in table order; its CUDA engine appends with one atomicAdd per triangle, so its output order changes
// from run to run.  Here the order is the CPU engine's, reproduced without atomics:
//   live list   the entries with ptr >= 0, ascending: one ordered selection over the scene's alloc_bits
//   count       one 512-thread workgroup per live block, one voxel per thread: cube case -> triangles per block
//   scan        exclusive scan of the per-block counts (one workgroup)
//   emit        same cube evaluation; an in-workgroup scan in voxel order gives every triangle its final slot
// so two runs give byte-identical meshes and the result can be compared bit for bit with the CPU oracle.
// The cube evaluation follows upstream's findPointNeighbors / buildVertList / sdfInterp: a cube is skipped when any
// of its 8 corner voxels is missing (block not allocated or swapped out) or has sdf == 1.
// Bound: gather latency (8 voxel reads per cube, 7 hash lookups per block); an offline export, not on the per-frame
// path -- no roofline claim.
#include <hip/hip_runtime.h>

#include "dslam_bits.h"
#include "mc_tables.h"

#pragma clang fp contract(off)

namespace dslam {

// device copy of the case table (16 B per case: one 128-bit load per cube)
__constant__ signed char d_mc_triangles[256][16];

struct MeshParams {
  const HashEntry *hash;
  const uint2 *voxels;
  int num_buckets;
  unsigned mask;
  const int *live_list;    // entry ids with ptr >= 0, ascending
  const int *live_count;
  int *block_counts;       // [live] triangles per block
  const int *block_offsets;  // [live] exclusive scan of block_counts
  float *positions;        // [limit][3][3]
  float *colours;          // [limit][3][3] or null
  float factor;            // voxel size: vertices leave in metres
  int limit;               // triangles with rank >= limit are dropped (upstream: noMaxTriangles - 1)
};

// the live list: every entry with a resident block, ascending -- the scene's alloc_bits as a list (dslam_bits.h)
struct SelLive {
  DSLAM_SEL_NO_LOAD
  const HashEntry *hash;
  __device__ bool test(int t, const NoPayload &) const { return hash[t].ptr >= 0; }
  __device__ void prologue() const {}
  __device__ int emit(int, int, bool, const NoPayload &) const { return 0; }
  __device__ void finish(int) const {}
};

// findVoxel's block search: walk the bucket's chain for a resident block at (bx, by, bz); -1 when there is none
__device__ __forceinline__ int find_block_ptr(const HashEntry *hash, int num_buckets, unsigned mask, int bx, int by, int bz) {
  int idx = hash_index(bx, by, bz, mask);
  while (true) {
    const HashEntry he = load_entry(hash, idx);
    if (he.pos[0] == bx && he.pos[1] == by && he.pos[2] == bz && he.ptr >= 0) return he.ptr;
    if (he.offset < 1) return -1;
    idx = num_buckets + he.offset - 1;
  }
}

// buildVertList's first half: the 8 corner samples and the case index; -1 = skip this cube.  With STORE the samples
// go to LDS ([corner][thread], conflict-free), where the emit pass indexes them by a run-time corner number.
template <bool STORE, bool COLOUR>
__device__ __forceinline__ int evaluate_cube(const MeshParams &p, const int *nb_ptr, int x, int y, int z,
                                             float (*s_sdf)[512], unsigned (*s_clr)[512]) {
  int cube = 0;
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int cx = x + mc_corner_x(k), cy = y + mc_corner_y(k), cz = z + mc_corner_z(k);
    const int nb = (cx >> 3) | ((cy >> 3) << 1) | ((cz >> 3) << 2);
    const int ptr = nb_ptr[nb];
    uint2 v = make_uint2(kEmptyVoxelLo, kEmptyVoxelHi);
    if (ptr >= 0) v = p.voxels[(size_t)ptr * kBlock3 + (cx & 7) + (cy & 7) * kBlock + (cz & 7) * kBlock * kBlock];
    const float s = sdf_to_float((short)(v.x & 0xffffu));
    // upstream returns at the first corner that fails; the outcome (skip) is the same whichever corner it is
    if (ptr < 0 || s == 1.0f) ok = false;
    if (STORE) s_sdf[k][threadIdx.x] = s;
    if (STORE && COLOUR) s_clr[k][threadIdx.x] = (v.x >> 24) | ((v.y & 0xffffu) << 8);  // r | g << 8 | b << 16
    if (s < 0.0f) cube |= 1 << k;
  }
  if (!ok || mc_edge_mask(cube) == 0) return -1;
  return cube;
}

__device__ __forceinline__ int triangles_of_case(int cube) {
  int n = 0;
#pragma unroll
  for (int i = 0; i < 15; i += 3) n += d_mc_triangles[cube][i] >= 0;
  return n;
}

// the neighbourhood of one block: its own slot and the 7 blocks at +x, +y, +z (threads 0..7 look them up)
__device__ __forceinline__ void resolve_neighbours(const MeshParams &p, const HashEntry &he, int *nb_ptr) {
  if (threadIdx.x < 8) {
    const int ox = threadIdx.x & 1, oy = (threadIdx.x >> 1) & 1, oz = threadIdx.x >> 2;
    nb_ptr[threadIdx.x] = threadIdx.x == 0 ? he.ptr
                                           : find_block_ptr(p.hash, p.num_buckets, p.mask, he.pos[0] + ox, he.pos[1] + oy, he.pos[2] + oz);
  }
  __syncthreads();
}

__global__ __launch_bounds__(512) void k_mesh_count(MeshParams p) {
  __shared__ int nb_ptr[8];
  __shared__ int red[8];
  const int live = *p.live_count;
  const int x = threadIdx.x & 7, y = (threadIdx.x >> 3) & 7, z = threadIdx.x >> 6;
  for (int b = blockIdx.x; b < live; b += gridDim.x) {
    const HashEntry he = load_entry(p.hash, p.live_list[b]);
    resolve_neighbours(p, he, nb_ptr);
    const int cube = evaluate_cube<false, false>(p, nb_ptr, x, y, z, nullptr, nullptr);
    int n = cube < 0 ? 0 : triangles_of_case(cube);
    for (int d = 32; d > 0; d >>= 1) n += __shfl_xor(n, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) p.block_counts[b] = red[0] + red[1] + red[2] + red[3] + red[4] + red[5] + red[6] + red[7];
    __syncthreads();
  }
}

__global__ __launch_bounds__(1024) void k_mesh_scan(const int *block_counts, int *block_offsets, const int *live_count,
                                                    int *total_out) {
  int totals[1];
  scan_tiles<1>(block_counts, block_offsets, *live_count, totals);
  if (threadIdx.x == 0) *total_out = totals[0];
}

// sdfInterp: the zero crossing between two corners (positions in voxel units); t is shared with the colour
struct Crossing { float t; int take; };  // take: 1 = first corner, 2 = second corner, 0 = interpolate with t
__device__ __forceinline__ Crossing crossing(float v1, float v2) {
  Crossing r{0.0f, 0};
  if (fabsf(0.0f - v1) < 0.00001f) r.take = 1;
  else if (fabsf(0.0f - v2) < 0.00001f) r.take = 2;
  else if (fabsf(v1 - v2) < 0.00001f) r.take = 1;
  else r.t = (0.0f - v1) / (v2 - v1);
  return r;
}
__device__ __forceinline__ float lerp_value(const Crossing &k, float a, float b) {
  if (k.take == 1) return a;
  if (k.take == 2) return b;
  return a + k.t * (b - a);
}

template <bool COLOUR>
__global__ __launch_bounds__(512) void k_mesh_emit(MeshParams p) {
  __shared__ int nb_ptr[8];
  __shared__ int wave_sums[8];
  __shared__ float s_sdf[8][512];
  __shared__ unsigned s_clr[COLOUR ? 8 : 1][512];
  const int live = *p.live_count;
  const int x = threadIdx.x & 7, y = (threadIdx.x >> 3) & 7, z = threadIdx.x >> 6;
  for (int b = blockIdx.x; b < live; b += gridDim.x) {
    if (p.block_counts[b] == 0) continue;  // uniform over the workgroup
    const HashEntry he = load_entry(p.hash, p.live_list[b]);
    resolve_neighbours(p, he, nb_ptr);
    const int cube = evaluate_cube<true, COLOUR>(p, nb_ptr, x, y, z, s_sdf, s_clr);
    const int n = cube < 0 ? 0 : triangles_of_case(cube);
    int total;
    int rank = block_excl_scan<8>(n, wave_sums, total) + p.block_offsets[b];
    if (n > 0) {
      const float gx = (float)(he.pos[0] * kBlock + x), gy = (float)(he.pos[1] * kBlock + y), gz = (float)(he.pos[2] * kBlock + z);
      for (int i = 0; i < 3 * n; i += 3, rank++) {
        if (rank >= p.limit) break;
        float *out = p.positions + (size_t)rank * 9;
        float *col = COLOUR ? p.colours + (size_t)rank * 9 : nullptr;
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const int e = d_mc_triangles[cube][i + k];
          const int a = mc_edge_first(e), bb = mc_edge_second(e);
          const Crossing cr = crossing(s_sdf[a][threadIdx.x], s_sdf[bb][threadIdx.x]);
          const float ax = gx + (float)mc_corner_x(a), ay = gy + (float)mc_corner_y(a), az = gz + (float)mc_corner_z(a);
          const float bx = gx + (float)mc_corner_x(bb), by = gy + (float)mc_corner_y(bb), bz = gz + (float)mc_corner_z(bb);
          out[3 * k + 0] = lerp_value(cr, ax, bx) * p.factor;
          out[3 * k + 1] = lerp_value(cr, ay, by) * p.factor;
          out[3 * k + 2] = lerp_value(cr, az, bz) * p.factor;
          if (COLOUR) {
            const unsigned ca = s_clr[a][threadIdx.x], cb = s_clr[bb][threadIdx.x];
            col[3 * k + 0] = lerp_value(cr, (float)(ca & 0xff), (float)(cb & 0xff)) / 255.0f;
            col[3 * k + 1] = lerp_value(cr, (float)((ca >> 8) & 0xff), (float)((cb >> 8) & 0xff)) / 255.0f;
            col[3 * k + 2] = lerp_value(cr, (float)((ca >> 16) & 0xff), (float)((cb >> 16) & 0xff)) / 255.0f;
          }
        }
      }
    }
    __syncthreads();  // nb_ptr / wave_sums are rewritten by the next block
  }
}

int launch_mesh_scene(dslam_engine *e, const dslam_scene *s, int max_triangles, int with_colour, int *out_num) {
  if (!e->mesh_table_ready) {  // __constant__ memory is per device: once per engine, not once per process
    DSLAM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(d_mc_triangles), kMcTriangles, sizeof(kMcTriangles)));
    e->mesh_table_ready = true;
  }
  const int N = s->n_entries;
  int rc = ensure_scratch(e, N, s->p.num_local_blocks);
  if (rc) return rc;
  int *live_count = e->misc_counter + 8, *total = e->misc_counter + 9;
  launch_bits_select(e, s->alloc_bits, N, SelLive{s->hash}, e->list_a, N, live_count, s->counters);
  MeshParams p;
  p.hash = s->hash; p.voxels = s->voxels; p.num_buckets = s->p.num_buckets; p.mask = (unsigned)(s->p.num_buckets - 1);
  p.live_list = e->list_a; p.live_count = live_count; p.block_counts = e->list_b; p.block_offsets = e->list_c;
  p.positions = nullptr; p.colours = nullptr; p.factor = s->p.voxel_size; p.limit = 0;
  const int grid = e->sm_count * 4;
  hipLaunchKernelGGL(k_mesh_count, dim3(grid), dim3(512), 0, e->stream, p);
  hipLaunchKernelGGL(k_mesh_scan, dim3(1), dim3(1024), 0, e->stream, e->list_b, e->list_c, live_count, total);
  DSLAM_HIP(hipGetLastError());
  int *host = reinterpret_cast<int *>(e->pinned);
  DSLAM_HIP(hipMemcpyAsync(host, total, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  DSLAM_HIP(hipStreamSynchronize(e->stream));
  // upstream: `triangles[n] = t; if (n < noMaxTriangles - 1) n++;` -- the list saturates at noMaxTriangles - 1
  const int n = host[0] < max_triangles - 1 ? host[0] : max_triangles - 1;
  const size_t need = (size_t)(n > 0 ? n : 1) * 9 * sizeof(float);
  if (need > e->mesh_bytes || (with_colour && !e->mesh_colours)) {
    if (e->mesh_positions) (void)hipFree(e->mesh_positions);
    if (e->mesh_colours) (void)hipFree(e->mesh_colours);
    e->mesh_positions = e->mesh_colours = nullptr;
    const size_t bytes = need > e->mesh_bytes ? need : e->mesh_bytes;
    e->mesh_bytes = 0;
    DSLAM_HIP(hipMalloc(&e->mesh_positions, bytes));
    if (with_colour) DSLAM_HIP(hipMalloc(&e->mesh_colours, bytes));
    e->mesh_bytes = bytes;
  }
  e->mesh_triangles = n;
  e->mesh_has_colour = with_colour != 0;
  if (n > 0) {
    p.positions = e->mesh_positions; p.colours = with_colour ? e->mesh_colours : nullptr; p.limit = n;
    if (with_colour) hipLaunchKernelGGL(k_mesh_emit<true>, dim3(grid), dim3(512), 0, e->stream, p);
    else hipLaunchKernelGGL(k_mesh_emit<false>, dim3(grid), dim3(512), 0, e->stream, p);
    DSLAM_HIP(hipGetLastError());
  }
  *out_num = n;
  return DSLAM_OK;
}

}  // namespace dslam
