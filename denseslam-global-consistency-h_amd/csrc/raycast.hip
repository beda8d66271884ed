// raycast.hip -- ITMVisualisationEngine for gfx950: FindVisibleBlocks, CountVisibleBlocks,
// CreateExpectedDepths, RenderImage (raycast + shading), CreateICPMaps.
//
// Reference call sites: ITMMainEngine::GetImage via InfiniTamDriver::GetImage / GetFloatImage
// (InfiniTamDriver.cpp:229-277, types :16-38), trackingController->Prepare (InfiniTamDriver.h:208-220),
// mapManager->countVisibleBlocks (DenseSlam.cpp:555-556).  Algorithm: SURVEY.md Appendix A.6, A.7.
//
// Mapping: the ray march is latency/gather bound (random 16-B hash probes + 8-B voxel reads).  One wavefront (= one
// workgroup) renders an 8x8 pixel tile, i.e. exactly one cell of the 1/8-resolution range image, so (zmin, zmax)
// and -- mostly -- the marched blocks are wave-uniform.
#include <cstdio>
#include <cstdlib>

#include "dslam_bits.h"

#pragma clang fp contract(off)

namespace dslam {

// ---------------------------------------------------------------------------------------------------------
// FindVisibleBlocks: ordered compaction of entries with ptr >= 0 that pass the 8-corner frustum test
// ---------------------------------------------------------------------------------------------------------
// An ordered selection over the scene's alloc_bits (dslam_bits.h): only entries that hold a block are read and tested
// (round 2: a frustum-flag sweep over all 1.18 M entries and a compaction sweep over 1.18 M byte flags).
struct FrustumParams {
  Mat4 M;
  float fx, fy, cx, cy, voxel_size;
  int W, H;
};

// ---------------------------------------------------------------------------------------------------------
// CountVisibleBlocks
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_count_visible(const int *__restrict__ ids, RenderCounters *rc,
                                                       const HashEntry *__restrict__ hash, int min_id, int max_id) {
  const int n = rc->no_visible;
  int c = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int ptr = hash[ids[i]].ptr;
    c += (ptr >= min_id && ptr <= max_id);
  }
  for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(&rc->count_result, c);
}

int launch_count_visible(dslam_engine *e, const dslam_scene *s, const dslam_render_state *r, int min_id, int max_id,
                         int *out) {
  DSLAM_HIP(hipMemsetAsync(&r->counters->count_result, 0, sizeof(int), e->stream));
  hipLaunchKernelGGL(k_count_visible, dim3(128), dim3(256), 0, e->stream, r->visible_ids, r->counters, s->hash, min_id,
                     max_id);
  int *host = reinterpret_cast<int *>(e->pinned);
  DSLAM_HIP(hipMemcpyAsync(host, &r->counters->count_result, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  DSLAM_HIP(hipStreamSynchronize(e->stream));
  *out = *host;
  return DSLAM_OK;
}

// ---------------------------------------------------------------------------------------------------------
// CreateExpectedDepths
// ---------------------------------------------------------------------------------------------------------
struct ProjParams {
  Mat4 M;
  float fx, fy, cx, cy, voxel_size;
  int W, H;
};

// ProjectSingleBlock: bbox (in 1/8-resolution cells) and z-range of one block; returns the number of 16x16 render
// tiles it needs (0 = nothing to render)
__device__ __forceinline__ int project_single_block(const HashEntry &e, const ProjParams &p, int4 &box, float2 &zr) {
  if (e.ptr < 0) return 0;
  int ulx = p.W / 8, uly = p.H / 8, lrx = -1, lry = -1;
  float zmin = kFarAway, zmax = kVeryClose;
#pragma unroll
  for (int corner = 0; corner < 8; corner++) {
    short tx = e.pos[0], ty = e.pos[1], tz = e.pos[2];
    tx += (corner & 1) ? 1 : 0; ty += (corner & 2) ? 1 : 0; tz += (corner & 4) ? 1 : 0;
    Vec4 q;
    q.x = (float)tx * (float)kBlock * p.voxel_size;
    q.y = (float)ty * (float)kBlock * p.voxel_size;
    q.z = (float)tz * (float)kBlock * p.voxel_size;
    q.w = 1.0f;
    q = mul(p.M, q);
    if (q.z < 1e-6f) continue;
    const float px = (p.fx * q.x / q.z + p.cx) / 8.0f;
    const float py = (p.fy * q.y / q.z + p.cy) / 8.0f;
    if ((float)ulx > floorf(px)) ulx = (int)floorf(px);
    if ((float)lrx < ceilf(px)) lrx = (int)ceilf(px);
    if ((float)uly > floorf(py)) uly = (int)floorf(py);
    if ((float)lry < ceilf(py)) lry = (int)ceilf(py);
    if (zmin > q.z) zmin = q.z;
    if (zmax < q.z) zmax = q.z;
  }
  if (ulx < 0) ulx = 0;
  if (uly < 0) uly = 0;
  if (lrx >= p.W) lrx = p.W - 1;
  if (lry >= p.H) lry = p.H - 1;
  bool valid = !(ulx > lrx) && !(uly > lry);
  if (zmin < kVeryClose) zmin = kVeryClose;
  if (zmax < kVeryClose) valid = false;
  if (!valid) return 0;
  const int rx = (int)ceilf((float)(lrx - ulx + 1) / 16.0f), ry = (int)ceilf((float)(lry - uly + 1) / 16.0f);
  box = make_int4(ulx, uly, lrx, lry);
  zr = make_float2(zmin, zmax);
  return rx * ry;
}

// PROJECT: the lane that lists visible entry number r also projects it (CreateExpectedDepths' ProjectSingleBlock; GetImage
// runs both with one pose), the compaction launch resets the range image, and every compaction tile leaves its
// render-tile total for k_fill_range_tiles.
template <bool PROJECT>
struct SelFrustum {
  const HashEntry *hash;
  FrustumParams fp;
  int4 *boxes;
  float2 *zr_out;
  int *req_out;
  float2 *range;
  int npix;
  __device__ void prologue() const {
    if (PROJECT)   // (independent job) reset the range image to (FAR_AWAY, VERY_CLOSE)
      for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) range[i] = make_float2(kFarAway, kVeryClose);
  }
  typedef HashEntry Payload;
  __device__ HashEntry load(int t) const { return load_entry(hash, t); }
  __device__ bool test(int, const HashEntry &e) const {
    if (e.ptr < 0) return false;
    bool vis, vis_enl;
    check_block_vis<false>(vis, vis_enl, e.pos[0], e.pos[1], e.pos[2], fp.M, fp.fx, fp.fy, fp.cx, fp.cy, fp.voxel_size, fp.W, fp.H);
    return vis;
  }
  struct Staged { int4 box; float2 zr; int req; };
  __device__ Staged stage(int, const HashEntry &e) const {
    Staged s;
    s.req = 0;
    if (!PROJECT) return s;
    ProjParams pp;
    pp.M = fp.M; pp.fx = fp.fx; pp.fy = fp.fy; pp.cx = fp.cx; pp.cy = fp.cy; pp.voxel_size = fp.voxel_size; pp.W = fp.W; pp.H = fp.H;
    s.req = project_single_block(e, pp, s.box, s.zr);
    return s;
  }
  __device__ int emit(int, int r, bool listed, const Staged &s) const {
    if (!PROJECT || !listed) return 0;
    if (s.req) { boxes[r] = s.box; zr_out[r] = s.zr; }
    req_out[r] = s.req;
    return s.req;
  }
  __device__ void finish(int) const {}
};

static FrustumParams make_frustum_params(const dslam_scene *s, const dslam_render_state *r, const float *M, const float *intr) {
  FrustumParams fp;
  memcpy(fp.M.m, M, 64);
  fp.fx = intr[0]; fp.fy = intr[1]; fp.cx = intr[2]; fp.cy = intr[3]; fp.voxel_size = s->p.voxel_size;
  fp.W = r->w; fp.H = r->h;
  return fp;
}

int launch_find_visible(dslam_engine *e, const dslam_scene *s, dslam_render_state *r, const float *M,
                        const float *intr) {
  const int N = s->n_entries;
  DSLAM_REQUIRE(r->n_entries == N, "render state was created for a different scene size");
  int rc = ensure_scratch(e, N, s->p.num_local_blocks);
  if (rc) return rc;
  SelFrustum<false> sel{s->hash, make_frustum_params(s, r, M, intr), nullptr, nullptr, nullptr, nullptr, 0};
  launch_bits_select(e, s->alloc_bits, N, sel, r->visible_ids, r->n_local, &r->counters->no_visible, s->counters);
  DSLAM_HIP(hipGetLastError());
  return DSLAM_OK;
}

// ProjectSingleBlock for every visible block; records bbox / z-range / required render tiles
__global__ __launch_bounds__(256) void k_project_blocks(const int *__restrict__ ids, RenderCounters *rc,
                                                        const HashEntry *__restrict__ hash, ProjParams p,
                                                        int4 *__restrict__ boxes, float2 *__restrict__ zr_out,
                                                        int *req_out, float2 *range, int npix, int *wg_tiles) {
  __shared__ int s_tiles;
  if (threadIdx.x == 0) s_tiles = 0;
  __syncthreads();
  // (independent job in the same launch) reset the range image to (FAR_AWAY, VERY_CLOSE)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x)
    range[i] = make_float2(kFarAway, kVeryClose);
  const int n = rc->no_visible;
  int local_tiles = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const HashEntry e = load_entry(hash, ids[i]);
    int4 box;
    float2 zr;
    const int req = project_single_block(e, p, box, zr);
    if (req) { boxes[i] = box; zr_out[i] = zr; }
    req_out[i] = req;
    local_tiles += req;
  }
  for (int d = 32; d > 0; d >>= 1) local_tiles += __shfl_down(local_tiles, d, 64);
  if ((threadIdx.x & 63) == 0 && local_tiles) atomicAdd(&s_tiles, local_tiles);
  __syncthreads();
  // per-workgroup totals instead of one contended global counter; the range-image kernel sums them
  if (threadIdx.x == 0) wg_tiles[blockIdx.x] = s_tiles;
}

// Fill the range image.  The render tiles of a block partition its bbox, so min/max over the bbox is identical to
// upstream's tile list.  Values are positive floats, so integer min/max on the bit patterns order correctly.
//
// Corner pass (the ceil(W/8) x ceil(H/8) cells the raycaster reads): one workgroup = one 16x16-cell tile x one chunk
// slice of the visible list.  Overlapping blocks are accumulated with LDS atomics (ds_min/ds_max), then every touched cell
// is flushed with ONE global atomic pair -- instead of one contended global atomic pair per (block, cell).
constexpr int kRangeTile = 16;
constexpr int kRangeSlices = 32;
#ifndef DSLAM_RANGE_BIG
#define DSLAM_RANGE_BIG 32
#endif
constexpr int kRangeBigBox = DSLAM_RANGE_BIG;   // cells of a tile above which a box is taken by the whole workgroup  // workgroups per tile; each strides over the visible list

__global__ __launch_bounds__(256) void k_fill_range_tiles(const RenderCounters *rc, const int4 *__restrict__ boxes,
                                                          const float2 *__restrict__ zr, const int *__restrict__ req,
                                                          float2 *range, int W, int tiles_x,
                                                          const int *__restrict__ wg_tiles, int n_wg_tiles, int budget,
                                                          int capacity) {
  __shared__ int s_min[kRangeTile * kRangeTile], s_max[kRangeTile * kRangeTile];
  __shared__ int red[4];
  // The visible count, the per-workgroup tile totals and this lane's first list entry are all fetched before anything
  // waits: four dependent round trips become one (the entry is read speculatively -- the index is inside the
  // buffers whatever the count turns out to be).
  const int i0 = blockIdx.y * 256 + threadIdx.x;
  int r0 = 0;
  int4 b0 = make_int4(0, 0, 0, 0);
  float2 z0 = make_float2(0.0f, 0.0f);
  if (i0 < capacity) { r0 = req[i0]; b0 = boxes[i0]; z0 = zr[i0]; }
  const int n = __builtin_amdgcn_readfirstlane(rc->no_visible);   // (uniform, and the compiler should know: a loop with barriers hangs off it)
  // The render-tile budget (MAX_RENDERING_BLOCKS) is applied in visible-list order; only when the total (the sum of
  // the projection pass' per-workgroup counts) exceeds it does the order matter.  That case (> 262144 tiles) is
  // replayed below, by every workgroup for itself.
  const bool over_budget = block_sum_strided(wg_tiles, n_wg_tiles, 1, red) >= budget;
  if ((int)(blockIdx.y * 256) >= n && !over_budget) return;
  const int tx0 = (blockIdx.x % tiles_x) * kRangeTile, ty0 = (blockIdx.x / tiles_x) * kRangeTile;
  const int far_i = __float_as_int(kFarAway), close_i = __float_as_int(kVeryClose);
  s_min[threadIdx.x] = far_i;
  s_max[threadIdx.x] = close_i;
  __syncthreads();
  auto splat_box = [&](const int4 &b, const float2 &z) {
    const int x0 = b.x > tx0 ? b.x : tx0, x1 = b.z < tx0 + kRangeTile - 1 ? b.z : tx0 + kRangeTile - 1;
    const int y0 = b.y > ty0 ? b.y : ty0, y1 = b.w < ty0 + kRangeTile - 1 ? b.w : ty0 + kRangeTile - 1;
    if (x0 > x1 || y0 > y1) return;
    const int zmin_i = __float_as_int(z.x), zmax_i = __float_as_int(z.y);
    for (int y = y0; y <= y1; y++)
      for (int x = x0; x <= x1; x++) {
        const int c = (x - tx0) + (y - ty0) * kRangeTile;
        atomicMin(&s_min[c], zmin_i);
        atomicMax(&s_max[c], zmax_i);
      }
  };
  auto splat = [&](int i) { splat_box(boxes[i], zr[i]); };
  if (!over_budget) {
    // A box that covers many cells of this tile (a block next to the camera: up to all 256) is not walked by the lane that
    // holds it -- 2 x 256 LDS atomics one after the other while the lanes with distant blocks are done after a handful --
    // but queued, and taken by the whole workgroup: thread c looks at cell c of the tile, in registers.
    __shared__ int4 s_big[256];
    __shared__ int2 s_bigz[256];
    __shared__ int s_nbig;
    int mn_c = far_i, mx_c = close_i;
    const int cx = tx0 + (threadIdx.x % kRangeTile), cy = ty0 + (threadIdx.x / kRangeTile);
    auto take = [&](bool have, const int4 &b, const float2 &z) {
      if (threadIdx.x == 0) s_nbig = 0;
      __syncthreads();
      if (have) {
        const int x0 = b.x > tx0 ? b.x : tx0, x1 = b.z < tx0 + kRangeTile - 1 ? b.z : tx0 + kRangeTile - 1;
        const int y0 = b.y > ty0 ? b.y : ty0, y1 = b.w < ty0 + kRangeTile - 1 ? b.w : ty0 + kRangeTile - 1;
        if (x0 <= x1 && y0 <= y1) {
          if ((x1 - x0 + 1) * (y1 - y0 + 1) > kRangeBigBox) {
            const int k = atomicAdd(&s_nbig, 1);
            s_big[k] = make_int4(x0, y0, x1, y1);
            s_bigz[k] = make_int2(__float_as_int(z.x), __float_as_int(z.y));
          } else {
            splat_box(b, z);
          }
        }
      }
      __syncthreads();
      const int nb = s_nbig;
      for (int k = 0; k < nb; k++) {
        const int4 bb = s_big[k];
        const int2 zz = s_bigz[k];
        if (cx >= bb.x && cx <= bb.z && cy >= bb.y && cy <= bb.w) {
          mn_c = zz.x < mn_c ? zz.x : mn_c;
          mx_c = zz.y > mx_c ? zz.y : mx_c;
        }
      }
      __syncthreads();   // (the queue is refilled by the next round)
    };
    take(i0 < n && r0 != 0, b0, z0);
    for (int ib = blockIdx.y * 256 + gridDim.y * 256; ib < n; ib += gridDim.y * 256) {   // (uniform trip count: barriers inside)
      const int i = ib + threadIdx.x;
      const bool have = i < n && req[i] != 0;
      int4 b = make_int4(0, 0, -1, -1);
      float2 z = make_float2(0.0f, 0.0f);
      if (have) { b = boxes[i]; z = zr[i]; }
      take(have, b, z);
    }
    if (mn_c != far_i) atomicMin(&s_min[threadIdx.x], mn_c);
    if (mx_c != close_i) atomicMax(&s_max[threadIdx.x], mx_c);
  } else {
    // sequential rule of the reference's tile list: entry i is dropped when the tiles accepted so far plus its own
    // reach the budget (a dropped entry does not count).  Chunks of the request list are staged in LDS, lane 0
    // replays them in order, then the lanes splat the accepted entries of this workgroup's slice.
    __shared__ int s_req[1024];
    __shared__ unsigned char s_keep[1024];
    __shared__ int s_num;
    if (threadIdx.x == 0) s_num = 0;
    for (int c0 = 0; c0 < n; c0 += 1024) {
      __syncthreads();
      for (int j = threadIdx.x; j < 1024; j += 256) s_req[j] = (c0 + j < n) ? req[c0 + j] : 0;
      __syncthreads();
      if (threadIdx.x == 0) {
        int num = s_num;
        for (int j = 0; j < 1024; j++) {
          const int r = s_req[j];
          bool keep = r != 0;
          if (keep) {
            if (num + r >= budget) keep = false;
            else num += r;
          }
          s_keep[j] = keep;
        }
        s_num = num;
      }
      __syncthreads();
      for (int j = threadIdx.x; j < 1024; j += 256) {
        const int i = c0 + j;
        // this workgroup's slice of the list, as in the common path: i = blockIdx.y * 256 + t (mod gridDim.y * 256)
        if (i < n && s_keep[j] && (i / 256) % (int)gridDim.y == (int)blockIdx.y) splat(i);
      }
    }
  }
  __syncthreads();
  const int mn = s_min[threadIdx.x], mx = s_max[threadIdx.x];
  if (mn != far_i || mx != close_i) {
    const int x = tx0 + (threadIdx.x % kRangeTile), y = ty0 + (threadIdx.x / kRangeTile);
    int *px = reinterpret_cast<int *>(&range[x + (size_t)y * W]);
    if (mn != far_i) atomicMin(&px[0], mn);
    if (mx != close_i) atomicMax(&px[1], mx);
  }
}

// Cells outside that corner: upstream clamps bboxes to the full image size although coordinates are 1/8 scale, so
// blocks near the camera spill thousands of cells past the corner.  Nothing ever reads them (castRay indexes
// floor(x/8) + floor(y/8) * W), so they are not filled here; tests compare the corner.

constexpr int kProjectGrid = 512;

static int launch_fill_range(dslam_engine *e, dslam_render_state *r, int n_wg_tiles) {
  // corner = the tiles covering ceil(W/8) x ceil(H/8) cells (clamped to the image); chunks sized for the pool
  const int cw = (r->w + 7) / 8, ch = (r->h + 7) / 8;
  const int tiles_x = (cw + kRangeTile - 1) / kRangeTile, tiles_y = (ch + kRangeTile - 1) / kRangeTile;
  hipLaunchKernelGGL(k_fill_range_tiles, dim3(tiles_x * tiles_y, kRangeSlices), dim3(256), 0, e->stream, r->counters,
                     r->proj_boxes, r->proj_z, r->proj_req, r->range, r->w, tiles_x, r->proj_wg_tiles, n_wg_tiles,
                     e->render_tile_budget, r->n_local);
  DSLAM_HIP(hipGetLastError());
  return DSLAM_OK;
}

static ProjParams make_proj_params(const dslam_scene *s, const dslam_render_state *r, const float *M, const float *intr) {
  ProjParams pp;
  memcpy(pp.M.m, M, 64);
  pp.fx = intr[0]; pp.fy = intr[1]; pp.cx = intr[2]; pp.cy = intr[3]; pp.voxel_size = s->p.voxel_size;
  pp.W = r->w; pp.H = r->h;
  return pp;
}

int launch_expected_depths(dslam_engine *e, const dslam_scene *s, dslam_render_state *r, const float *M,
                           const float *intr) {
  const ProjParams pp = make_proj_params(s, r, M, intr);
  hipLaunchKernelGGL(k_project_blocks, dim3(kProjectGrid), dim3(256), 0, e->stream, r->visible_ids, r->counters, s->hash,
                     pp, r->proj_boxes, r->proj_z, r->proj_req, r->range, r->w * r->h, r->proj_wg_tiles);
  return launch_fill_range(e, r, kProjectGrid);
}

// FindVisibleBlocks + CreateExpectedDepths for the same pose (ITMMainEngine::GetImage's FREECAMERA path): three launches
// (round 1: five), none of which reads the table as a whole any more.
int launch_find_visible_and_depths(dslam_engine *e, const dslam_scene *s, dslam_render_state *r, const float *M,
                                   const float *intr) {
  const int N = s->n_entries;
  DSLAM_REQUIRE(r->n_entries == N, "render state was created for a different scene size");
  int rc = ensure_scratch(e, N, s->p.num_local_blocks);
  if (rc) return rc;
  SelFrustum<true> sel{s->hash, make_frustum_params(s, r, M, intr), r->proj_boxes, r->proj_z, r->proj_req, r->range, r->w * r->h};
  launch_bits_select(e, s->alloc_bits, N, sel, r->visible_ids, r->n_local, &r->counters->no_visible, s->counters, r->proj_wg_tiles);
  return launch_fill_range(e, r, select_tiles(N));
}

// ---------------------------------------------------------------------------------------------------------
// voxel access
// ---------------------------------------------------------------------------------------------------------
struct VolumeRef {
  const HashEntry *hash;
  const uint2 *voxels;
  unsigned mask;
  int num_buckets;
};

struct IndexCache {
  int bx, by, bz, block_ptr;
};

__device__ __forceinline__ uint2 read_voxel(const VolumeRef &vol, int px, int py, int pz, bool &found, IndexCache &c) {
  const int bx = ((px < 0) ? px - kBlock + 1 : px) / kBlock;
  const int by = ((py < 0) ? py - kBlock + 1 : py) / kBlock;
  const int bz = ((pz < 0) ? pz - kBlock + 1 : pz) / kBlock;
  const int lin = (px - bx * kBlock) + (py - by * kBlock) * kBlock + (pz - bz * kBlock) * kBlock * kBlock;
  if (bx == c.bx && by == c.by && bz == c.bz) {
    found = true;
    return vol.voxels[(size_t)c.block_ptr + lin];
  }
  int h = hash_index(bx, by, bz, vol.mask);
  while (true) {
    const HashEntry e = load_entry(vol.hash, h);
    if (e.pos[0] == bx && e.pos[1] == by && e.pos[2] == bz && e.ptr >= 0) {
      found = true;
      c.bx = bx; c.by = by; c.bz = bz;
      c.block_ptr = e.ptr * kBlock3;
      return vol.voxels[(size_t)c.block_ptr + lin];
    }
    if (e.offset < 1) break;
    h = vol.num_buckets + e.offset - 1;
  }
  found = false;
  return make_uint2(kEmptyVoxelLo, kEmptyVoxelHi);
}

__device__ __forceinline__ float rd_sdf(const VolumeRef &vol, int x, int y, int z, bool &found, IndexCache &c) {
  return (float)(short)(read_voxel(vol, x, y, z, found, c).x & 0xffffu);
}

// (int)(x < 0 ? x - 0.5f : x + 0.5f); copysign folds the compare + select into one bit-field insert (-0.0 gives 0
// either way)
__device__ __forceinline__ int iround(float x) { return (int)(x + __builtin_copysignf(0.5f, x)); }

__device__ __forceinline__ float read_sdf_uninterp(const VolumeRef &vol, const Vec3 &pt, bool &found, IndexCache &c) {
  return rd_sdf(vol, iround(pt.x), iround(pt.y), iround(pt.z), found, c) / 32767.0f;
}

// block base pointer (voxel index of the block's first voxel) or -1; refreshes the per-lane cache on a hit
__device__ __forceinline__ int lookup_block(const VolumeRef &vol, int bx, int by, int bz, IndexCache &c) {
  if (bx == c.bx && by == c.by && bz == c.bz) return c.block_ptr;
  int h = hash_index(bx, by, bz, vol.mask);
  while (true) {
    const HashEntry e = load_entry(vol.hash, h);
    if (e.pos[0] == bx && e.pos[1] == by && e.pos[2] == bz && e.ptr >= 0) {
      c.bx = bx; c.by = by; c.bz = bz;
      c.block_ptr = e.ptr * kBlock3;
      return c.block_ptr;
    }
    if (e.offset < 1) return -1;
    h = vol.num_buckets + e.offset - 1;
  }
}

// Block base pointers (voxel index of the block's first voxel, or -1) of the 8 corners of a trilinear cell whose
// per-axis block coordinates are bxa/bya/bza[0..1]; corner k = (k & 1, (k >> 1) & 1, k >> 2).  All eight bucket heads
// are loaded in ONE round trip (equal blocks hit the same address); corners whose head holds another block follow
// their excess chains together, one round trip per link.
__device__ __forceinline__ void resolve_cell_blocks(const VolumeRef &vol, const int bxa[2], const int bya[2],
                                                    const int bza[2], int base[8]) {
  const unsigned hx[2] = {(unsigned)bxa[0] * 73856093u, (unsigned)bxa[1] * 73856093u};
  const unsigned hy[2] = {(unsigned)bya[0] * 19349669u, (unsigned)bya[1] * 19349669u};
  const unsigned hz[2] = {(unsigned)bza[0] * 83492791u, (unsigned)bza[1] * 83492791u};
  // packed position words of an entry: x = pos0 | pos1 << 16, y = pos2 (| pad); a block coordinate outside the
  // short range can never be stored, so it never matches
  const unsigned tx[2] = {(unsigned)bxa[0] & 0xffffu, (unsigned)bxa[1] & 0xffffu};
  const unsigned ty[2] = {(unsigned)bya[0] << 16, (unsigned)bya[1] << 16};
  const unsigned tz[2] = {(unsigned)bza[0] & 0xffffu, (unsigned)bza[1] & 0xffffu};
  const bool okx[2] = {bxa[0] == (short)bxa[0], bxa[1] == (short)bxa[1]};
  const bool oky[2] = {bya[0] == (short)bya[0], bya[1] == (short)bya[1]};
  const bool okz[2] = {bza[0] == (short)bza[0], bza[1] == (short)bza[1]};
  int h[8];
  u32x4 e[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    h[k] = (int)((hx[k & 1] ^ hy[(k >> 1) & 1] ^ hz[k >> 2]) & vol.mask);
    e[k] = *reinterpret_cast<const u32x4 *>(vol.hash + h[k]);
  }
  unsigned pending = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const bool ok = okx[k & 1] && oky[(k >> 1) & 1] && okz[k >> 2];
    const bool match = ok && e[k].x == (tx[k & 1] | ty[(k >> 1) & 1]) && (e[k].y & 0xffffu) == tz[k >> 2] && (int)e[k].w >= 0;
    base[k] = match ? (int)e[k].w * kBlock3 : -1;
    if (!match && (int)e[k].z >= 1) { pending |= 1u << k; h[k] = vol.num_buckets + (int)e[k].z - 1; }
  }
  // excess chains: all unresolved corners advance one link per round trip.  Branch-free on purpose -- a wave64
  // executes every instruction any of its lanes needs, and eight predicated blocks with a branch each cost five
  // times the instructions of selects (resolved corners just re-read their last entry and ignore it).
  while (pending) {
#pragma unroll
    for (int k = 0; k < 8; k++) e[k] = *reinterpret_cast<const u32x4 *>(vol.hash + h[k]);
    unsigned still = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const bool pk = (pending & (1u << k)) != 0;
      const bool ok = okx[k & 1] && oky[(k >> 1) & 1] && okz[k >> 2];
      const bool match = ok && e[k].x == (tx[k & 1] | ty[(k >> 1) & 1]) && (e[k].y & 0xffffu) == tz[k >> 2] && (int)e[k].w >= 0;
      const bool cont = pk && !match && (int)e[k].z >= 1;
      base[k] = (pk && match) ? (int)e[k].w * kBlock3 : base[k];
      h[k] = cont ? vol.num_buckets + (int)e[k].z - 1 : h[k];
      still |= cont ? (1u << k) : 0u;
    }
    pending = still;
  }
}

// The 8 taps of a trilinear read.  Tap k = (dx, dy, dz) = (k & 1, (k >> 1) & 1, k >> 2) relative to (x, y, z).  The
// kernel is bound by the NUMBER of scattered load instructions (measured: prefetching or speculative variants that
// add loads are slower), so the <= 8 (usually 1 or 2) distinct voxel blocks are resolved with as few probes as
// possible -- de-duplicated per axis, per-lane block cache first -- and then the 8 voxel loads are issued together.
// A tap whose block is not allocated reads the empty voxel, exactly like readVoxel.
__device__ __forceinline__ void gather_taps(const VolumeRef &vol, int x, int y, int z, IndexCache &c, uint2 t[8]) {
  const int bx0 = x >> 3, by0 = y >> 3, bz0 = z >> 3;  // arithmetic shift = floor division, as pointToVoxelBlockPos
  const int bx1 = (x + 1) >> 3, by1 = (y + 1) >> 3, bz1 = (z + 1) >> 3;
  const bool sx = bx1 == bx0, sy = by1 == by0, sz = bz1 == bz0;
  int p[8];
  p[0] = lookup_block(vol, bx0, by0, bz0, c);
  p[1] = sx ? p[0] : lookup_block(vol, bx1, by0, bz0, c);
  p[2] = sy ? p[0] : lookup_block(vol, bx0, by1, bz0, c);
  p[3] = sx ? p[2] : (sy ? p[1] : lookup_block(vol, bx1, by1, bz0, c));
  p[4] = sz ? p[0] : lookup_block(vol, bx0, by0, bz1, c);
  p[5] = sz ? p[1] : (sx ? p[4] : lookup_block(vol, bx1, by0, bz1, c));
  p[6] = sz ? p[2] : (sy ? p[4] : lookup_block(vol, bx0, by1, bz1, c));
  p[7] = sz ? p[3] : (sx ? p[6] : (sy ? p[5] : lookup_block(vol, bx1, by1, bz1, c)));
  const int lx0 = x & 7, lx1 = (x + 1) & 7, ly0 = (y & 7) * kBlock, ly1 = ((y + 1) & 7) * kBlock;
  const int lz0 = (z & 7) * kBlock * kBlock, lz1 = ((z + 1) & 7) * kBlock * kBlock;
  const int lin[8] = {lx0 + ly0 + lz0, lx1 + ly0 + lz0, lx0 + ly1 + lz0, lx1 + ly1 + lz0,
                      lx0 + ly0 + lz1, lx1 + ly0 + lz1, lx0 + ly1 + lz1, lx1 + ly1 + lz1};
#pragma unroll
  for (int k = 0; k < 8; k++) {
    // clamp the address so the load is unconditional (and therefore batched); select afterwards
    const uint2 v = vol.voxels[(size_t)(p[k] >= 0 ? p[k] : 0) + lin[k]];
    t[k] = (p[k] >= 0) ? v : make_uint2(kEmptyVoxelLo, kEmptyVoxelHi);
  }
}

// a / b via the correctly rounded reciprocal y = RN(1/b): exactly RN(a/b) for b = 32767 (0 mismatches over every
// finite float a, tests/tools/verify_exact_div.cpp); 3 instructions instead of the ~10 of an IEEE division
__device__ __forceinline__ float div_exact(float a, float b, float y) {
  const float q = a * y;
  const float r = __fmaf_rn(-b, q, a);
  return __fmaf_rn(r, y, q);
}

__device__ __forceinline__ float trilinear_sdf(const uint2 t[8], float cx, float cy, float cz) {
  float s[8];
#pragma unroll
  for (int k = 0; k < 8; k++) s[k] = (float)(short)(t[k].x & 0xffffu);
  float res1 = (1.0f - cx) * s[0] + cx * s[1];
  res1 = (1.0f - cy) * res1 + cy * ((1.0f - cx) * s[2] + cx * s[3]);
  float res2 = (1.0f - cx) * s[4] + cx * s[5];
  res2 = (1.0f - cy) * res2 + cy * ((1.0f - cx) * s[6] + cx * s[7]);
  return div_exact((1.0f - cz) * res1 + cz * res2, 32767.0f, 1.0f / 32767.0f);
}

// The 8 taps (low voxel words) of the trilinear cell at (x0, y0, z0) in two load round trips: every block of the
// cell resolved together, then the 8 taps together; a tap whose block is not allocated reads the empty voxel.
__device__ __forceinline__ void gather_taps_batched(const VolumeRef &vol, int x0, int y0, int z0, unsigned raw[8]) {
  const int bxa[2] = {x0 >> 3, (x0 + 1) >> 3}, bya[2] = {y0 >> 3, (y0 + 1) >> 3}, bza[2] = {z0 >> 3, (z0 + 1) >> 3};
  int base[8];
  resolve_cell_blocks(vol, bxa, bya, bza, base);
  const unsigned lx[2] = {(unsigned)x0 & 7u, (unsigned)(x0 + 1) & 7u};
  const unsigned ly[2] = {((unsigned)y0 & 7u) << 3, ((unsigned)(y0 + 1) & 7u) << 3};
  const unsigned lz[2] = {((unsigned)z0 & 7u) << 6, ((unsigned)(z0 + 1) & 7u) << 6};
  const char *vbytes = reinterpret_cast<const char *>(vol.voxels);
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const unsigned lin = lx[k & 1] | ly[(k >> 1) & 1] | lz[k >> 2];
    const unsigned off = (base[k] < 0) ? 0u : ((unsigned)base[k] + lin) * 8u;
    raw[k] = *reinterpret_cast<const unsigned *>(vbytes + off);
  }
#pragma unroll
  for (int k = 0; k < 8; k++) raw[k] = (base[k] < 0) ? kEmptyVoxelLo : raw[k];
}

__device__ __forceinline__ float trilinear_raw(const unsigned raw[8], float cx, float cy, float cz) {
  uint2 t[8];
#pragma unroll
  for (int k = 0; k < 8; k++) t[k] = make_uint2(raw[k], 0u);
  return trilinear_sdf(t, cx, cy, cz);
}

// readFromSDF_float_interpolated; same values as read_sdf_interp, two round trips instead of up to nine
__device__ __forceinline__ float read_sdf_interp_batched(const VolumeRef &vol, const Vec3 &pt) {
  const float fx = floorf(pt.x), fy = floorf(pt.y), fz = floorf(pt.z);
  unsigned raw[8];
  gather_taps_batched(vol, (int)fx, (int)fy, (int)fz, raw);
  return trilinear_raw(raw, pt.x - fx, pt.y - fy, pt.z - fz);
}

__device__ __forceinline__ float read_sdf_interp(const VolumeRef &vol, const Vec3 &pt, bool &found, IndexCache &c) {
  const float fx = floorf(pt.x), fy = floorf(pt.y), fz = floorf(pt.z);
  const int x = (int)fx, y = (int)fy, z = (int)fz;
  const float cx = pt.x - fx, cy = pt.y - fy, cz = pt.z - fz;
  uint2 t[8];
  gather_taps(vol, x, y, z, c, t);
  found = true;
  return trilinear_sdf(t, cx, cy, cz);
}

__device__ __forceinline__ Vec4 read_colour_interp(const VolumeRef &vol, const Vec3 &pt, IndexCache &c) {
  const float fx = floorf(pt.x), fy = floorf(pt.y), fz = floorf(pt.z);
  const int x = (int)fx, y = (int)fy, z = (int)fz;
  const float cx = pt.x - fx, cy = pt.y - fy, cz = pt.z - fz;
  float rx = 0.0f, ry = 0.0f, rz = 0.0f;
  uint2 t[8];
  gather_taps(vol, x, y, z, c, t);
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int ox = k & 1, oy = (k >> 1) & 1, oz = (k >> 2) & 1;
    const uint2 v = t[k];
    const float wx = ox ? cx : (1.0f - cx), wy = oy ? cy : (1.0f - cy), wz = oz ? cz : (1.0f - cz);
    const float w = wx * wy * wz;
    rx += w * (float)(v.x >> 24);
    ry += w * (float)(v.y & 0xffu);
    rz += w * (float)((v.y >> 8) & 0xffu);
  }
  Vec4 r = {rx / 255.0f, ry / 255.0f, rz / 255.0f, 255.0f / 255.0f};
  return r;
}

// computeSingleNormalFromSDF (un-normalised gradient)
__device__ __forceinline__ Vec3 normal_from_sdf(const VolumeRef &vol, const Vec3 &pt, IndexCache &c) {
  bool f;
  Vec3 ret;
  const float flx = floorf(pt.x), fly = floorf(pt.y), flz = floorf(pt.z);
  const int x = (int)flx, y = (int)fly, z = (int)flz;
  const float cx = pt.x - flx, cy = pt.y - fly, cz = pt.z - flz;
  const float nx = 1.0f - cx, ny = 1.0f - cy, nz = 1.0f - cz;
  Vec4 front, back, tmp;
  front.x = rd_sdf(vol, x, y, z, f, c); front.y = rd_sdf(vol, x + 1, y, z, f, c);
  front.z = rd_sdf(vol, x, y + 1, z, f, c); front.w = rd_sdf(vol, x + 1, y + 1, z, f, c);
  back.x = rd_sdf(vol, x, y, z + 1, f, c); back.y = rd_sdf(vol, x + 1, y, z + 1, f, c);
  back.z = rd_sdf(vol, x, y + 1, z + 1, f, c); back.w = rd_sdf(vol, x + 1, y + 1, z + 1, f, c);
  float p1, p2, v1;
  // gradient x
  p1 = front.x * ny * nz + front.z * cy * nz + back.x * ny * cz + back.z * cy * cz;
  tmp.x = rd_sdf(vol, x - 1, y, z, f, c); tmp.y = rd_sdf(vol, x - 1, y + 1, z, f, c);
  tmp.z = rd_sdf(vol, x - 1, y, z + 1, f, c); tmp.w = rd_sdf(vol, x - 1, y + 1, z + 1, f, c);
  p2 = tmp.x * ny * nz + tmp.y * cy * nz + tmp.z * ny * cz + tmp.w * cy * cz;
  v1 = p1 * cx + p2 * nx;
  p1 = front.y * ny * nz + front.w * cy * nz + back.y * ny * cz + back.w * cy * cz;
  tmp.x = rd_sdf(vol, x + 2, y, z, f, c); tmp.y = rd_sdf(vol, x + 2, y + 1, z, f, c);
  tmp.z = rd_sdf(vol, x + 2, y, z + 1, f, c); tmp.w = rd_sdf(vol, x + 2, y + 1, z + 1, f, c);
  p2 = tmp.x * ny * nz + tmp.y * cy * nz + tmp.z * ny * cz + tmp.w * cy * cz;
  ret.x = (p1 * nx + p2 * cx - v1) / 32767.0f;
  // gradient y
  p1 = front.x * nx * nz + front.y * cx * nz + back.x * nx * cz + back.y * cx * cz;
  tmp.x = rd_sdf(vol, x, y - 1, z, f, c); tmp.y = rd_sdf(vol, x + 1, y - 1, z, f, c);
  tmp.z = rd_sdf(vol, x, y - 1, z + 1, f, c); tmp.w = rd_sdf(vol, x + 1, y - 1, z + 1, f, c);
  p2 = tmp.x * nx * nz + tmp.y * cx * nz + tmp.z * nx * cz + tmp.w * cx * cz;
  v1 = p1 * cy + p2 * ny;
  p1 = front.z * nx * nz + front.w * cx * nz + back.z * nx * cz + back.w * cx * cz;
  tmp.x = rd_sdf(vol, x, y + 2, z, f, c); tmp.y = rd_sdf(vol, x + 1, y + 2, z, f, c);
  tmp.z = rd_sdf(vol, x, y + 2, z + 1, f, c); tmp.w = rd_sdf(vol, x + 1, y + 2, z + 1, f, c);
  p2 = tmp.x * nx * nz + tmp.y * cx * nz + tmp.z * nx * cz + tmp.w * cx * cz;
  ret.y = (p1 * ny + p2 * cy - v1) / 32767.0f;
  // gradient z
  p1 = front.x * nx * ny + front.y * cx * ny + front.z * nx * cy + front.w * cx * cy;
  tmp.x = rd_sdf(vol, x, y, z - 1, f, c); tmp.y = rd_sdf(vol, x + 1, y, z - 1, f, c);
  tmp.z = rd_sdf(vol, x, y + 1, z - 1, f, c); tmp.w = rd_sdf(vol, x + 1, y + 1, z - 1, f, c);
  p2 = tmp.x * nx * ny + tmp.y * cx * ny + tmp.z * nx * cy + tmp.w * cx * cy;
  v1 = p1 * cz + p2 * nz;
  p1 = back.x * nx * ny + back.y * cx * ny + back.z * nx * cy + back.w * cx * cy;
  tmp.x = rd_sdf(vol, x, y, z + 2, f, c); tmp.y = rd_sdf(vol, x + 1, y, z + 2, f, c);
  tmp.z = rd_sdf(vol, x, y + 1, z + 2, f, c); tmp.w = rd_sdf(vol, x + 1, y + 1, z + 2, f, c);
  p2 = tmp.x * nx * ny + tmp.y * cx * ny + tmp.z * nx * cy + tmp.w * cx * cy;
  ret.z = (p1 * nz + p2 * cz - v1) / 32767.0f;
  return ret;
}

// ---------------------------------------------------------------------------------------------------------
// castRay + shading, one kernel
// ---------------------------------------------------------------------------------------------------------
struct RenderParams {
  VolumeRef vol;
  Mat4 M, invM;
  float inv_fx, inv_fy, cx, cy;
  float one_over_vs, voxel_size, mu, inv_32767;
  int W, H;
  const float2 *range;
  float4 *raycast;
  uchar4 *out_rgba;
  float *out_float;
  int type;  // dslam_image_type, or -1: raycast only
  unsigned long long *dbg_waves;  // diagnostics only (env DSLAM_DBG_WAVETIME=<file>): per wave {cycles, max iterations, straddling iterations, their cycles, setup cycles, refinement cycles}
  int dbg_flags;  // diagnostics only (env DSLAM_DBG_FLAGS: 8 = 16x16 workgroups)
  float split_len;  // > 0: tiles with a longer depth range (voxels) are marched by two wavefronts; the grid is (W/8, 2 H/8)
};

constexpr float kSplitLen = 175.0f;  // voxels of depth range above which a tile is marched by two wavefronts (150-200 measure the same)

// DIAG instantiation only: wave-level split of the march (single-wave workgroups): iterations in which some lane took
// the straddling-cell path, and the cycles of those iterations
struct MarchDiag { int iters, wave_iters, slow_iters; unsigned long long slow_cycles, setup_cycles, tail_cycles; };

template <bool DIAG>
__device__ __forceinline__ bool cast_ray(Vec4 &out, int x, int y, const RenderParams &p, const float2 minmax,
                                         MarchDiag &diag) {
  const unsigned long long t_enter = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
  Vec4 pc;
  Vec3 ps, pe, dir, res;
  float sdf = 1.0f;
  float total, step, total_max;
  const float step_scale = p.mu * p.one_over_vs;

  pc.z = minmax.x;
  pc.x = pc.z * (((float)x - p.cx) * p.inv_fx);
  pc.y = pc.z * (((float)y - p.cy) * p.inv_fy);
  pc.w = 1.0f;
  total = sqrtf(pc.x * pc.x + pc.y * pc.y + pc.z * pc.z) * p.one_over_vs;
  Vec4 q = mul(p.invM, pc);
  ps.x = q.x * p.one_over_vs; ps.y = q.y * p.one_over_vs; ps.z = q.z * p.one_over_vs;

  pc.z = minmax.y;
  pc.x = pc.z * (((float)x - p.cx) * p.inv_fx);
  pc.y = pc.z * (((float)y - p.cy) * p.inv_fy);
  pc.w = 1.0f;
  total_max = sqrtf(pc.x * pc.x + pc.y * pc.y + pc.z * pc.z) * p.one_over_vs;
  q = mul(p.invM, pc);
  pe.x = q.x * p.one_over_vs; pe.y = q.y * p.one_over_vs; pe.z = q.z * p.one_over_vs;

  dir.x = pe.x - ps.x; dir.y = pe.y - ps.y; dir.z = pe.z - ps.z;
  const float dn = 1.0f / sqrtf(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
  dir.x *= dn; dir.y *= dn; dir.z *= dn;
  res = ps;
  IndexCache cache = {0x7fffffff, 0x7fffffff, 0x7fffffff, -1};
  int iter = 0;
  __shared__ int s_slow_flag;
  unsigned long long t_iter = 0;
  if (DIAG) {
    diag.slow_iters = 0; diag.wave_iters = 0; diag.slow_cycles = 0;
    diag.setup_cycles = __builtin_amdgcn_s_memtime() - t_enter;
    s_slow_flag = 0;
    t_iter = __builtin_amdgcn_s_memtime();
  }
  while (total < total_max) {
    ++iter;
    if (DIAG) {  // account the previous wave iteration
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      if (iter > 1) {
        diag.wave_iters++;
        if (s_slow_flag) { diag.slow_iters++; diag.slow_cycles += now - t_iter; }
      }
      s_slow_flag = 0;
      t_iter = now;
    }
    // Measured on MI355X (DSLAM_DBG_WAVETIME dump): the launch keeps only ~1.1 waves per SIMD resident on average
    // and ends when its longest wave ends, and a lone wave64 issues one VALU instruction per 4 cycles -- a step costs
    // its load round trips (~700 cycles each) PLUS 4 cycles for every instruction ANY lane of the wave executes.
    // Hence: (a) the common step is kept short -- one probe for the block of the nearest voxel ROUND(p), then the
    // 8 taps of the trilinear cell floor(p)..floor(p)+1 in one batch from THAT block (in-block offsets wrap, so the
    // loads are unconditional; all 8 taps lie in it at 67 % of the positions, ROUND(p) always does); (b) the
    // uncommon step -- near the surface with the cell straddling blocks -- resolves all its blocks in one further
    // round trip and re-reads the taps in a second (read_sdf_interp_batched), instead of readFromSDF_float_
    // interpolated's eight lookups one after the other.  Variants that add loads or instructions to the common
    // step (bitmap, prefetch, speculation, resolving the whole cell every step) all measured slower.
    const int vx = iround(res.x), vy = iround(res.y), vz = iround(res.z);
    const int bx = vx >> 3, by = vy >> 3, bz = vz >> 3;
    const int base = lookup_block(p.vol, bx, by, bz, cache);
    if (base < 0) {
      sdf = 1.0f;  // empty voxel: 32767 / 32767
      step = (float)kBlock;
    } else {
      const float f0x = floorf(res.x), f0y = floorf(res.y), f0z = floorf(res.z);
      const int x0 = (int)f0x, y0 = (int)f0y, z0 = (int)f0z;
      unsigned raw[8];
      // The 8 taps as 32-bit byte offsets from the (uniform) voxel array: one or3 + one shift-add per tap and the
      // load takes the scalar base.  A tap that belongs to a neighbouring block still reads a valid address inside
      // this block; its value is only used when all 8 taps are inside (all_in).
      const unsigned lx[2] = {(unsigned)x0 & 7u, (unsigned)(x0 + 1) & 7u};
      const unsigned ly[2] = {((unsigned)y0 & 7u) << 3, ((unsigned)(y0 + 1) & 7u) << 3};
      const unsigned lz[2] = {((unsigned)z0 & 7u) << 6, ((unsigned)(z0 + 1) & 7u) << 6};
      const char *vbytes = reinterpret_cast<const char *>(p.vol.voxels);
      const unsigned off0 = (unsigned)base * 8u;  // base < 2^27 voxels -> < 2^30 bytes
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const unsigned off = off0 + ((lx[k & 1] | ly[(k >> 1) & 1] | lz[k >> 2]) << 3);
        raw[k] = *reinterpret_cast<const unsigned *>(vbytes + off);  // unconditional, batched
      }
      // nearest voxel ROUND(p) = tap (vx - x0, vy - y0, vz - z0): a 3-level select tree
      const bool nx = vx != x0, ny = vy != y0, nz = vz != z0;
      const unsigned r01 = nx ? raw[1] : raw[0], r23 = nx ? raw[3] : raw[2];
      const unsigned r45 = nx ? raw[5] : raw[4], r67 = nx ? raw[7] : raw[6];
      const unsigned rn = nz ? (ny ? r67 : r45) : (ny ? r23 : r01);
      sdf = div_exact((float)(short)(rn & 0xffffu), 32767.0f, p.inv_32767);
      if ((sdf <= 0.1f) && (sdf >= -0.5f)) {
        const bool all_in = (x0 >> 3) == bx && ((x0 + 1) >> 3) == bx && (y0 >> 3) == by && ((y0 + 1) >> 3) == by &&
                            (z0 >> 3) == bz && ((z0 + 1) >> 3) == bz;
        if (!all_in) {  // the cell straddles blocks: fetch it properly
          if (DIAG) s_slow_flag = 1;
          gather_taps_batched(p.vol, x0, y0, z0, raw);
        }
        sdf = trilinear_raw(raw, res.x - f0x, res.y - f0y, res.z - f0z);
      }
      if (sdf <= 0.0f) break;
      step = fmaxf(sdf * step_scale, 1.0f);
    }
    res.x += step * dir.x; res.y += step * dir.y; res.z += step * dir.z;
    total += step;
  }
  diag.iters = iter;
  const unsigned long long t_tail = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
  bool pt_found;
  if (sdf <= 0.0f) {
    step = sdf * step_scale;
    res.x += step * dir.x; res.y += step * dir.y; res.z += step * dir.z;
    sdf = read_sdf_interp_batched(p.vol, res);
    step = sdf * step_scale;
    res.x += step * dir.x; res.y += step * dir.y; res.z += step * dir.z;
    pt_found = true;
  } else {
    pt_found = false;
  }
  out.x = res.x; out.y = res.y; out.z = res.z; out.w = pt_found ? 1.0f : 0.0f;
  if (DIAG) diag.tail_cycles = __builtin_amdgcn_s_memtime() - t_tail;
  return pt_found;
}

// SHADE = false: raycast only / depth image -- the variant the fusion loop and the tracker use; it carries no
// shading code, which keeps it at <= 64 VGPRs = 8 waves per SIMD (the march is latency-bound, occupancy is what
// hides its load round trips).  SHADE = true adds the normal / colour modes.
// REUSE = true: raycastResult already holds this very view's march (GetImage memo) -- shade only.
template <int WAVES, bool SHADE, bool DIAG = false, bool REUSE = false>
__global__ __launch_bounds__(WAVES * 64, 5) void k_render(RenderParams p) {
  // one wavefront = one workgroup = an 8x8 pixel tile = exactly one cell of the 1/8-resolution range image.
  // Single-wave workgroups let the dispatcher backfill a SIMD the moment a short tile finishes (ray lengths vary
  // by 10x between tiles), instead of holding 4 waves until the slowest of a 16x16 tile is done.
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // (measured: dealing each XCD a contiguous band of tiles for L2 locality is slower, 104 vs 100 us -- the long
  // rays of one image region then pile up on one XCD; the march is bound by its longest dependent-load chain)
  const int x = (WAVES == 4) ? blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7) : blockIdx.x * 8 + (lane & 7);
  int y = (WAVES == 4) ? blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3) : blockIdx.y * 8 + (lane >> 3);
  if (WAVES == 1 && p.split_len > 0.0f) {
    // The launch ends when its longest wavefront ends, and a step of a wavefront costs the union of what its rays do
    // (measured: the first 32 steps of the longest tile, all 64 rays alive, take twice as long as its last 36).  So a
    // tile whose rays have far to go -- depth range of its cell of the range image, in voxels -- is marched by TWO
    // wavefronts of 32 rays (upper / lower four rows): 78 -> 70 us on the bench scene.  Splitting every tile loses
    // (throughput: 9600 half-empty waves), 16 rays per wave loses; the grid has two workgroups per tile and the second
    // one of an unsplit tile leaves at once.  Rays are independent: the images do not change.
    const int ty = blockIdx.y >> 1, sub = blockIdx.y & 1;
    const float2 mm = p.range[(int)blockIdx.x + ty * p.W];
    const bool split = (mm.y - mm.x) * p.one_over_vs > p.split_len;
    if (split ? (lane >= 32) : (sub != 0)) return;
    y = ty * 8 + (split ? sub * 4 : 0) + (lane >> 3);
    // (the launch ends with these wavefronts: while short tiles share their SIMDs they issue first -- 138.2 -> 137.5 us per frame,
    // four alternations, every run of the one below every run of the other)
    if (split) __builtin_amdgcn_s_setprio(3);   // (a second level -- tiles half as long at priority 2 -- measures the same)
  }
  if (x >= p.W || y >= p.H) return;
  const int loc = x + y * p.W;
  const int loc2 = (int)floorf((float)x / 8.0f) + (int)floorf((float)y / 8.0f) * p.W;
  Vec4 pr;
  MarchDiag diag;
  const unsigned long long t_start = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
  if (REUSE) {
    const float4 q = p.raycast[loc];
    pr.x = q.x; pr.y = q.y; pr.z = q.z; pr.w = q.w;
  } else {
    cast_ray<DIAG>(pr, x, y, p, p.range[loc2], diag);
  }
  if (DIAG) {  // diagnostic instantiation (DSLAM_DBG_WAVETIME): per-wave cycles and march length
    const unsigned long long dt = __builtin_amdgcn_s_memtime() - t_start;
    // the lane with the longest march saw every wave iteration
    int mx = diag.iters;
    for (int d = 32; d > 0; d >>= 1) { const int o = __shfl_xor(mx, d, 64); mx = mx > o ? mx : o; }
    if (diag.iters == mx) {  // (several lanes may tie; they write the same values)
      const int wid = (blockIdx.y * gridDim.x + blockIdx.x) * WAVES + wave;
      p.dbg_waves[6 * wid] = dt;
      p.dbg_waves[6 * wid + 1] = (unsigned long long)mx;
      p.dbg_waves[6 * wid + 2] = (unsigned long long)diag.slow_iters;
      p.dbg_waves[6 * wid + 3] = diag.slow_cycles;
      p.dbg_waves[6 * wid + 4] = diag.setup_cycles;
      p.dbg_waves[6 * wid + 5] = diag.tail_cycles;
    }
  }
  if (!REUSE) p.raycast[loc] = make_float4(pr.x, pr.y, pr.z, pr.w);
  if (p.type < 0) return;

  const Vec3 pt = {pr.x, pr.y, pr.z};
  bool found = pr.w > 0;
  if (p.type == DSLAM_IMAGE_DEPTH) {
    float d = 0.0f;
    if (found) {
      Vec4 pw = {pt.x * p.voxel_size, pt.y * p.voxel_size, pt.z * p.voxel_size, 1.0f};
      d = mul(p.M, pw).z;
    }
    p.out_float[loc] = d;
    return;
  }
  if (!SHADE) return;
  IndexCache c = {0x7fffffff, 0x7fffffff, 0x7fffffff, -1};
  Vec3 n = {0, 0, 0};
  float angle = 0.0f;
  if (found) {
    const Vec3 light = {-p.invM.m[8], -p.invM.m[9], -p.invM.m[10]};
    n = normal_from_sdf(p.vol, pt, c);
    const float ns = 1.0f / sqrtf(n.x * n.x + n.y * n.y + n.z * n.z);
    n.x *= ns; n.y *= ns; n.z *= ns;
    angle = n.x * light.x + n.y * light.y + n.z * light.z;
    if (!(angle > 0.0f)) found = false;
  }
  uchar4 o = make_uchar4(0, 0, 0, 0);
  if (found) {
    if (p.type == DSLAM_IMAGE_COLOUR_FROM_VOLUME) {
      const Vec4 clr = read_colour_interp(p.vol, pt, c);
      o = make_uchar4((unsigned char)(clr.x * 255.0f), (unsigned char)(clr.y * 255.0f), (unsigned char)(clr.z * 255.0f),
                      255);
    } else if (p.type == DSLAM_IMAGE_COLOUR_FROM_NORMAL) {
      o = make_uchar4((unsigned char)((0.3f + (-n.x + 1.0f) * 0.35f) * 255.0f),
                      (unsigned char)((0.3f + (-n.y + 1.0f) * 0.35f) * 255.0f),
                      (unsigned char)((0.3f + (-n.z + 1.0f) * 0.35f) * 255.0f), 255);
    } else {
      const unsigned char g = (unsigned char)((0.8f * angle + 0.2f) * 255.0f);
      o = make_uchar4(g, g, g, g);
    }
  }
  p.out_rgba[loc] = o;
}

static int fill_render_params(RenderParams &rp, const dslam_scene *s, dslam_render_state *r, const float *M,
                              const float *intr, int type) {
  rp.vol.hash = s->hash; rp.vol.voxels = s->voxels; rp.vol.mask = (unsigned)(s->p.num_buckets - 1);
  rp.vol.num_buckets = s->p.num_buckets;
  memcpy(rp.M.m, M, 64);
  if (!invert_matrix(M, rp.invM.m)) { set_last_error("pose matrix is singular"); return DSLAM_ERR_INVALID; }
  rp.inv_fx = 1.0f / intr[0]; rp.inv_fy = 1.0f / intr[1]; rp.cx = intr[2]; rp.cy = intr[3];
  rp.voxel_size = s->p.voxel_size; rp.one_over_vs = 1.0f / s->p.voxel_size; rp.mu = s->p.mu;
  rp.inv_32767 = 1.0f / 32767.0f;
  rp.W = r->w; rp.H = r->h;
  rp.range = r->range; rp.raycast = r->raycast; rp.out_rgba = r->image_rgba; rp.out_float = r->image_float;
  rp.type = type;
  static const int dbg_flags = getenv("DSLAM_DBG_FLAGS") ? atoi(getenv("DSLAM_DBG_FLAGS")) : 0;
  rp.dbg_flags = dbg_flags; rp.dbg_waves = nullptr; rp.split_len = 0.0f;
  return DSLAM_OK;
}

// grid of the single-wave march kernels; `split`: long tiles get two wavefronts (see k_render)
static dim3 march_grid(RenderParams &rp, const dslam_render_state *r, bool split) {
  static const float split_len = getenv("DSLAM_RENDER_SPLIT") ? (float)atof(getenv("DSLAM_RENDER_SPLIT")) : kSplitLen;
  rp.split_len = split ? split_len : 0.0f;
  return dim3((r->w + 7) / 8, ((r->h + 7) / 8) * (rp.split_len > 0.0f ? 2 : 1));
}

int launch_render(dslam_engine *e, const dslam_scene *s, dslam_render_state *r, const float *M, const float *intr,
                  int type, bool reuse_raycast, void *image_out_override) {
  RenderParams rp;
  int rc = fill_render_params(rp, s, r, M, intr, type);
  if (rc) return rc;
  if (image_out_override) {  // a page-locked caller image: the kernel stores the pixels there itself (over PCIe)
    if (type == DSLAM_IMAGE_DEPTH) rp.out_float = static_cast<float *>(image_out_override);
    else rp.out_rgba = static_cast<uchar4 *>(image_out_override);
  }
  static const char *dbg_file = getenv("DSLAM_DBG_WAVETIME");
  static int dbg_calls = 0;
  const int n_waves = ((r->w + 7) / 8) * ((r->h + 7) / 8) * 2;
  unsigned long long *dbg_host = nullptr;
  if (dbg_file && ++dbg_calls == 30) {  // one snapshot, well into the run
    DSLAM_HIP(hipHostMalloc((void **)&dbg_host, (size_t)n_waves * 48, hipHostMallocDefault));
    memset(dbg_host, 0, (size_t)n_waves * 48);
    rp.dbg_waves = dbg_host;
  }
  const dim3 grid1 = march_grid(rp, r, !reuse_raycast);
  if (reuse_raycast) {
    if (type == DSLAM_IMAGE_DEPTH || type < 0) hipLaunchKernelGGL((k_render<1, false, false, true>), grid1, dim3(64), 0, e->stream, rp);
    else hipLaunchKernelGGL((k_render<1, true, false, true>), grid1, dim3(64), 0, e->stream, rp);
  } else if (rp.dbg_flags & 8)
    hipLaunchKernelGGL((k_render<4, true>), dim3((r->w + 15) / 16, (r->h + 15) / 16), dim3(256), 0, e->stream, rp);
  else if (rp.dbg_waves)
    hipLaunchKernelGGL((k_render<1, false, true>), grid1, dim3(64), 0, e->stream, rp);
  else if (type == DSLAM_IMAGE_DEPTH || type < 0)
    hipLaunchKernelGGL((k_render<1, false>), grid1, dim3(64), 0, e->stream, rp);
  else
    hipLaunchKernelGGL((k_render<1, true>), grid1, dim3(64), 0, e->stream, rp);
  DSLAM_HIP(hipGetLastError());
  if (dbg_host) {
    DSLAM_HIP(hipStreamSynchronize(e->stream));
    FILE *f = fopen(dbg_file, "wb");
    if (f) { fwrite(dbg_host, 48, n_waves, f); fclose(f); }
    (void)hipHostFree(dbg_host);
  }
  return DSLAM_OK;
}

// ---------------------------------------------------------------------------------------------------------
// CreateICPMaps: processPixelICP<useSmoothing = true, flipNormals = false>
// ---------------------------------------------------------------------------------------------------------
// Also drawPixelGrey into renderState->raycastImage: the picture ITMMainEngine::GetImage(InfiniTAM_IMAGE_SCENERAYCAST)
// copies out (PreviewType::kRaycastImage, InfiniTamDriver.cpp:28-29) -- (0.8 angle + 0.2) 255 on all four channels, 0
// where the ray found nothing.
__global__ __launch_bounds__(256) void k_icp_maps(const float4 *__restrict__ pr, int W, int H, float vs, float lx,
                                                  float ly, float lz, float4 *points, float4 *normals,
                                                  uchar4 *grey) {
  const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (x >= W || y >= H) return;
  const int loc = x + y * W;
  const float4 point = pr[loc];
  bool found = point.w > 0.0f;
  float nx = 0, ny = 0, nz = 0, angle = 0.0f;
  if (found && (y <= 2 || y >= H - 3 || x <= 2 || x >= W - 3)) found = false;
  if (found) {
    float4 xp = pr[(x + 2) + y * W], yp = pr[x + (y + 2) * W], xm = pr[(x - 2) + y * W], ym = pr[x + (y - 2) * W];
    float4 dx = make_float4(0, 0, 0, 0), dy = make_float4(0, 0, 0, 0);
    bool plus1 = false;
    if (xp.w <= 0 || yp.w <= 0 || xm.w <= 0 || ym.w <= 0) plus1 = true;
    if (!plus1) {
      dx = make_float4(xp.x - xm.x, xp.y - xm.y, xp.z - xm.z, xp.w - xm.w);
      dy = make_float4(yp.x - ym.x, yp.y - ym.y, yp.z - ym.z, yp.w - ym.w);
      const float ld = fmaxf(dx.x * dx.x + dx.y * dx.y + dx.z * dx.z, dy.x * dy.x + dy.y * dy.y + dy.z * dy.z);
      if (ld * vs * vs > (0.15f * 0.15f)) plus1 = true;
    }
    if (plus1) {
      xp = pr[(x + 1) + y * W]; yp = pr[x + (y + 1) * W]; xm = pr[(x - 1) + y * W]; ym = pr[x + (y - 1) * W];
      dx = make_float4(xp.x - xm.x, xp.y - xm.y, xp.z - xm.z, xp.w - xm.w);
      dy = make_float4(yp.x - ym.x, yp.y - ym.y, yp.z - ym.z, yp.w - ym.w);
      if (xp.w <= 0 || yp.w <= 0 || xm.w <= 0 || ym.w <= 0) found = false;
    }
    if (found) {
      nx = -(dx.y * dy.z - dx.z * dy.y);
      ny = -(dx.z * dy.x - dx.x * dy.z);
      nz = -(dx.x * dy.y - dx.y * dy.x);
      const float ns = 1.0f / sqrtf(nx * nx + ny * ny + nz * nz);
      nx *= ns; ny *= ns; nz *= ns;
      angle = nx * lx + ny * ly + nz * lz;
      if (!(angle > 0.0f)) found = false;
    }
  }
  if (found) {
    const unsigned char g = (unsigned char)((0.8f * angle + 0.2f) * 255.0f);
    grey[loc] = make_uchar4(g, g, g, g);
    points[loc] = make_float4(point.x * vs, point.y * vs, point.z * vs, 1.0f);
    normals[loc] = make_float4(nx, ny, nz, 0.0f);
  } else {
    grey[loc] = make_uchar4(0, 0, 0, 0);
    points[loc] = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
    normals[loc] = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
  }
}

int launch_icp_maps(dslam_engine *e, const dslam_scene *s, dslam_render_state *r, const float *M, const float *intr) {
  if (!r->icp_points) {
    DSLAM_HIP(hipMalloc(&r->icp_points, (size_t)r->w * r->h * sizeof(float4)));
    DSLAM_HIP(hipMalloc(&r->icp_normals, (size_t)r->w * r->h * sizeof(float4)));
    DSLAM_HIP(hipMalloc(&r->raycast_image, (size_t)r->w * r->h * sizeof(uchar4)));
  }
  RenderParams rp;
  int rc = fill_render_params(rp, s, r, M, intr, -1);
  if (rc) return rc;
  const dim3 grid((r->w + 15) / 16, (r->h + 15) / 16);
  hipLaunchKernelGGL((k_render<1, false>), march_grid(rp, r, true), dim3(64), 0, e->stream, rp);
  hipLaunchKernelGGL(k_icp_maps, grid, dim3(256), 0, e->stream, r->raycast, r->w, r->h, s->p.voxel_size, -rp.invM.m[8],
                     -rp.invM.m[9], -rp.invM.m[10], r->icp_points, r->icp_normals, r->raycast_image);
  DSLAM_HIP(hipGetLastError());
  return DSLAM_OK;
}

}  // namespace dslam
