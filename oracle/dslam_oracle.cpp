/*
 * dslam_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the voxel-block-hashing TSDF hot path that Hansry/DenseSLAM-Global-Consistency-h
 * reaches through ITMLib (allocate -> integrate -> raycast, de-integration, voxel decay, sliding window,
 * swap in/out).  It is the checker for libdslam_fusion.so: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product never does.
 *
 * PARITY UNPINNED.  The implementing module of the reference is an un-vendored git submodule
 *   src/InfiniTAM-Global-Consistency-h -> https://github.com/Hansry/InfiniTAM-Global-Consistency-h.git
 * (/root/reference/.gitmodules:7-9; directory empty, pinned commit unknown) and the reference ships no
 * tests, golden vectors or fixtures for this path.  What follows restates the published
 * algorithm of the code that fork is built on -- InfiniTAM v2 (victorprad/InfiniTAM, ITMLib/Engine/...,
 * DeviceAgnostic/ITMSceneReconstructionEngine.h, ITMRepresentationAccess.h, ITMVisualisationEngine.h,
 * ITMSwappingEngine.h) plus DynSLAM's voxel decay -- as recorded in SURVEY.md Appendix A, anchored on the
 * reference's own call sites.  Each function cites the reference call site (file:line under
 * /root/reference/src/DenseSLAM) and the Appendix A section it follows.  Where the fork's behaviour is not
 * knowable the choice made here is stated; those are design decisions, not parity claims.
 *
 * Build: see oracle/Makefile (g++ -O2 -ffp-contract=off, optional -fopenmp).  All float arithmetic is
 * written in the operation order of the upstream code so the HIP kernels can match it bit for bit.
 */
#include "../include/dslam_fusion.h"
// the marching-cubes case table is plain data shared with the engine; tests/test_mc_tables.py validates every row
#include "../denseslam-global-consistency-h_amd/csrc/mc_tables.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ------------------------------------------------------------------------------------------------
// small vector / matrix helpers (ORUtils::Vector*/Matrix4f restated; column-major m[16])
// ------------------------------------------------------------------------------------------------
struct V2f { float x, y; };
struct V2i { int x, y; };
struct V3f { float x, y, z; };
struct V3i { int x, y, z; };
struct V4f { float x, y, z, w; };
struct S4 { int16_t x, y, z, w; };

// Matrix4f * Vector4f, ORUtils operator order: m[0]*x + m[4]*y + m[8]*z + m[12]*w
static inline V4f mul(const float *m, const V4f &v) {
  V4f r;
  r.x = m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12] * v.w;
  r.y = m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13] * v.w;
  r.z = m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14] * v.w;
  r.w = m[3] * v.x + m[7] * v.y + m[11] * v.z + m[15] * v.w;
  return r;
}

// ORUtils::Matrix4::inv (classic cofactor expansion on the transposed source).  ITMPose::GetInvM()
// (InfiniTamDriver.h:153,159,176) is this function applied to M.
static bool inv4(const float *m, float *dst) {
  float tmp[12], src[16], det;
  for (int i = 0; i < 4; i++) {
    src[i] = m[i * 4];
    src[i + 4] = m[i * 4 + 1];
    src[i + 8] = m[i * 4 + 2];
    src[i + 12] = m[i * 4 + 3];
  }
  tmp[0] = src[10] * src[15];
  tmp[1] = src[11] * src[14];
  tmp[2] = src[9] * src[15];
  tmp[3] = src[11] * src[13];
  tmp[4] = src[9] * src[14];
  tmp[5] = src[10] * src[13];
  tmp[6] = src[8] * src[15];
  tmp[7] = src[11] * src[12];
  tmp[8] = src[8] * src[14];
  tmp[9] = src[10] * src[12];
  tmp[10] = src[8] * src[13];
  tmp[11] = src[9] * src[12];

  dst[0] = (tmp[0] * src[5] + tmp[3] * src[6] + tmp[4] * src[7]) - (tmp[1] * src[5] + tmp[2] * src[6] + tmp[5] * src[7]);
  dst[1] = (tmp[1] * src[4] + tmp[6] * src[6] + tmp[9] * src[7]) - (tmp[0] * src[4] + tmp[7] * src[6] + tmp[8] * src[7]);
  dst[2] = (tmp[2] * src[4] + tmp[7] * src[5] + tmp[10] * src[7]) - (tmp[3] * src[4] + tmp[6] * src[5] + tmp[11] * src[7]);
  dst[3] = (tmp[5] * src[4] + tmp[8] * src[5] + tmp[11] * src[6]) - (tmp[4] * src[4] + tmp[9] * src[5] + tmp[10] * src[6]);
  dst[4] = (tmp[1] * src[1] + tmp[2] * src[2] + tmp[5] * src[3]) - (tmp[0] * src[1] + tmp[3] * src[2] + tmp[4] * src[3]);
  dst[5] = (tmp[0] * src[0] + tmp[7] * src[2] + tmp[8] * src[3]) - (tmp[1] * src[0] + tmp[6] * src[2] + tmp[9] * src[3]);
  dst[6] = (tmp[3] * src[0] + tmp[6] * src[1] + tmp[11] * src[3]) - (tmp[2] * src[0] + tmp[7] * src[1] + tmp[10] * src[3]);
  dst[7] = (tmp[4] * src[0] + tmp[9] * src[1] + tmp[10] * src[2]) - (tmp[5] * src[0] + tmp[8] * src[1] + tmp[11] * src[2]);

  tmp[0] = src[2] * src[7];
  tmp[1] = src[3] * src[6];
  tmp[2] = src[1] * src[7];
  tmp[3] = src[3] * src[5];
  tmp[4] = src[1] * src[6];
  tmp[5] = src[2] * src[5];
  tmp[6] = src[0] * src[7];
  tmp[7] = src[3] * src[4];
  tmp[8] = src[0] * src[6];
  tmp[9] = src[2] * src[4];
  tmp[10] = src[0] * src[5];
  tmp[11] = src[1] * src[4];

  dst[8] = (tmp[0] * src[13] + tmp[3] * src[14] + tmp[4] * src[15]) - (tmp[1] * src[13] + tmp[2] * src[14] + tmp[5] * src[15]);
  dst[9] = (tmp[1] * src[12] + tmp[6] * src[14] + tmp[9] * src[15]) - (tmp[0] * src[12] + tmp[7] * src[14] + tmp[8] * src[15]);
  dst[10] = (tmp[2] * src[12] + tmp[7] * src[13] + tmp[10] * src[15]) - (tmp[3] * src[12] + tmp[6] * src[13] + tmp[11] * src[15]);
  dst[11] = (tmp[5] * src[12] + tmp[8] * src[13] + tmp[11] * src[14]) - (tmp[4] * src[12] + tmp[9] * src[13] + tmp[10] * src[14]);
  dst[12] = (tmp[2] * src[10] + tmp[5] * src[11] + tmp[1] * src[9]) - (tmp[4] * src[11] + tmp[0] * src[9] + tmp[3] * src[10]);
  dst[13] = (tmp[8] * src[11] + tmp[0] * src[8] + tmp[7] * src[10]) - (tmp[6] * src[10] + tmp[9] * src[11] + tmp[1] * src[8]);
  dst[14] = (tmp[6] * src[9] + tmp[11] * src[11] + tmp[3] * src[8]) - (tmp[10] * src[11] + tmp[2] * src[8] + tmp[7] * src[9]);
  dst[15] = (tmp[10] * src[10] + tmp[4] * src[8] + tmp[9] * src[9]) - (tmp[8] * src[9] + tmp[11] * src[10] + tmp[5] * src[8]);

  det = src[0] * dst[0] + src[1] * dst[1] + src[2] * dst[2] + src[3] * dst[3];
  if (det == 0.0f) {
    for (int i = 0; i < 16; i++) dst[i] = 0.0f;
    return false;
  }
  for (int i = 0; i < 16; i++) dst[i] = dst[i] * (1.0f / det);
  return true;
}

// ------------------------------------------------------------------------------------------------
// objects
// ------------------------------------------------------------------------------------------------
static const dslam_voxel kEmptyVoxel = {32767, 0, {0, 0, 0}, 0, 0};

}  // namespace

struct oracle_engine {
  dslam_weight_params wp;
  int threads;
  int render_tile_budget = DSLAM_MAX_RENDERING_BLOCKS;
  std::vector<float> mesh_pos, mesh_col;  // the last mesh oracle_mesh_scene produced ([n][3][3] each)
  bool mesh_has_colour = false;
};

struct oracle_scene {
  dslam_scene_params p;
  int n_entries;
  std::vector<dslam_hash_entry> hash;
  std::vector<dslam_voxel> vba;
  std::vector<int32_t> alloc_list, excess_list;
  int last_free, last_free_ex;
  std::vector<uint8_t> alloc_type;  // entriesAllocType scratch
  std::vector<S4> block_coords;     // blockCoords scratch
  // Visible-list history (DESIGN.md "visible-list rings"): instead of DynSLAM's queue of per-frame id
  // lists, every voxel-block slot carries two bit rings (0 fusion, 1 defusion); list k of ring q owns
  // bit k % (64*history_words).  A block is referenced by a queued list iff its bit is set.
  int history_words;
  std::vector<uint64_t> masks;      // [slot][ring][word]
  int ring_head[2], ring_next[2];   // oldest live list index, next list index
  int decay_cursor[2];              // next list index the aged-list Decay has to process
  std::vector<int32_t> last_seen;   // per slot: global frame counter of the newest list holding it;
                                    // -1 never; <= -2: (-2 - idx) = seen at idx, already swept by Decay
  int frame_counter;
  int64_t decayed_blocks, slid_blocks;
  int alloc_failures, last_swapped_in, last_swapped_out;
  // ITMGlobalCache
  std::vector<uint8_t> swap_state;
  std::vector<uint8_t> has_stored;
  dslam_voxel *stored;  // n_entries * 512, calloc'ed lazily by the OS
  int shard, num_shards, chunk_blocks;
  int shard_first, shard_count;
  // sharded re-integration: per slot "a (de-)integration pass visited this block since tracking began"
  std::vector<uint8_t> dirty;
  bool dirty_tracking = false;
  std::vector<int32_t> dirty_list;      // dirty slots, shard by shard, ascending inside a shard
  std::vector<int32_t> dirty_counts;    // per shard
  int dirty_chunk = 0;
};

struct oracle_render_state {
  int w, h;
  int n_entries, n_local;
  std::vector<int32_t> visible_ids;
  int no_visible;
  std::vector<uint8_t> visible_type;
  std::vector<V2f> range;    // renderingRangeImage, full image stride
  std::vector<V4f> raycast;  // raycastResult
  std::vector<uint8_t> image_rgba;
  std::vector<float> image_float;
  std::vector<V4f> icp_points, icp_normals;  // CreateICPMaps output, kept for the depth tracker
  std::vector<uint8_t> raycast_image;        // ITMRenderState::raycastImage: CreateICPMaps' grey rendering (drawPixelGrey)
};

struct oracle_view {
  int w_rgb, h_rgb, w_d, h_d;
  std::vector<uint8_t> rgba;
  std::vector<float> depth;
  std::vector<float> float_image;  // ITMViewBuilder::floatImage (bilateral filter scratch)
  std::vector<int16_t> raw_depth;  // the millimetre image of the last update (copied into the keyframe store)
  double timestamp;
};

// The visible list has room for num_local_blocks entries (visibleEntryIDs is sized SDF_LOCAL_BLOCK_NUM upstream).
// With swapping, entries parked on the host are visible too, so on a tiny pool the list can fill up: entries past
// the capacity are dropped, in ascending entry order -- the same clipping the HIP engine's compaction applies.
static inline void push_visible(oracle_render_state *r, int &n, int t) {
  if (n < r->n_local) r->visible_ids[n] = t;
  n++;
}

namespace {

// ------------------------------------------------------------------------------------------------
// hashing / addressing (KATs in Appendix C)
// ------------------------------------------------------------------------------------------------
static inline int hash_index(int bx, int by, int bz, uint32_t mask) {
  return (int)((((uint32_t)bx * 73856093u) ^ ((uint32_t)by * 19349669u) ^ ((uint32_t)bz * 83492791u)) & mask);
}

static inline int point_to_block(const V3i &p, V3i &b) {
  b.x = ((p.x < 0) ? p.x - DSLAM_BLOCK_SIZE + 1 : p.x) / DSLAM_BLOCK_SIZE;
  b.y = ((p.y < 0) ? p.y - DSLAM_BLOCK_SIZE + 1 : p.y) / DSLAM_BLOCK_SIZE;
  b.z = ((p.z < 0) ? p.z - DSLAM_BLOCK_SIZE + 1 : p.z) / DSLAM_BLOCK_SIZE;
  int lx = p.x - b.x * DSLAM_BLOCK_SIZE, ly = p.y - b.y * DSLAM_BLOCK_SIZE, lz = p.z - b.z * DSLAM_BLOCK_SIZE;
  return lx + ly * DSLAM_BLOCK_SIZE + lz * DSLAM_BLOCK_SIZE * DSLAM_BLOCK_SIZE;
}

struct IndexCache {
  V3i block_pos;
  int block_ptr;
  IndexCache() : block_pos{INT_MAX, INT_MAX, INT_MAX}, block_ptr(-1) {}
};

// readVoxel
static inline dslam_voxel read_voxel(const oracle_scene *s, const V3i &p, bool &found, IndexCache &cache) {
  V3i b;
  int lin = point_to_block(p, b);
  if (b.x == cache.block_pos.x && b.y == cache.block_pos.y && b.z == cache.block_pos.z) {
    found = true;
    return s->vba[(size_t)cache.block_ptr + lin];
  }
  int h = hash_index(b.x, b.y, b.z, (uint32_t)(s->p.num_buckets - 1));
  while (true) {
    const dslam_hash_entry &e = s->hash[h];
    if (e.pos[0] == b.x && e.pos[1] == b.y && e.pos[2] == b.z && e.ptr >= 0) {
      found = true;
      cache.block_pos = b;
      cache.block_ptr = e.ptr * DSLAM_BLOCK_SIZE3;
      return s->vba[(size_t)cache.block_ptr + lin];
    }
    if (e.offset < 1) break;
    h = s->p.num_buckets + e.offset - 1;
  }
  found = false;
  return kEmptyVoxel;
}

static inline float sdf_to_float(int16_t v) { return (float)v / 32767.0f; }
static inline int16_t float_to_sdf(float x) { return (int16_t)(x * 32767.0f); }

static inline int iround(float x) { return (int)((x < 0) ? (x - 0.5f) : (x + 0.5f)); }

static inline float read_sdf_uninterp(const oracle_scene *s, const V3f &pt, bool &found, IndexCache &c) {
  V3i p = {iround(pt.x), iround(pt.y), iround(pt.z)};
  dslam_voxel v = read_voxel(s, p, found, c);
  return sdf_to_float(v.sdf);
}

static inline void floor3(const V3f &pt, V3i &pos, V3f &coeff) {
  float fx = floorf(pt.x), fy = floorf(pt.y), fz = floorf(pt.z);
  pos.x = (int)fx; pos.y = (int)fy; pos.z = (int)fz;
  coeff.x = pt.x - fx; coeff.y = pt.y - fy; coeff.z = pt.z - fz;
}

static inline float rd(const oracle_scene *s, const V3i &pos, int dx, int dy, int dz, bool &found, IndexCache &c) {
  V3i q = {pos.x + dx, pos.y + dy, pos.z + dz};
  return (float)read_voxel(s, q, found, c).sdf;
}

// readFromSDF_float_interpolated (SURVEY A.7 read_trilinear)
static inline float read_sdf_interp(const oracle_scene *s, const V3f &pt, bool &found, IndexCache &c) {
  float res1, res2, v1, v2;
  V3f coeff; V3i pos;
  floor3(pt, pos, coeff);
  v1 = rd(s, pos, 0, 0, 0, found, c); v2 = rd(s, pos, 1, 0, 0, found, c);
  res1 = (1.0f - coeff.x) * v1 + coeff.x * v2;
  v1 = rd(s, pos, 0, 1, 0, found, c); v2 = rd(s, pos, 1, 1, 0, found, c);
  res1 = (1.0f - coeff.y) * res1 + coeff.y * ((1.0f - coeff.x) * v1 + coeff.x * v2);
  v1 = rd(s, pos, 0, 0, 1, found, c); v2 = rd(s, pos, 1, 0, 1, found, c);
  res2 = (1.0f - coeff.x) * v1 + coeff.x * v2;
  v1 = rd(s, pos, 0, 1, 1, found, c); v2 = rd(s, pos, 1, 1, 1, found, c);
  res2 = (1.0f - coeff.y) * res2 + coeff.y * ((1.0f - coeff.x) * v1 + coeff.x * v2);
  found = true;
  return ((1.0f - coeff.z) * res1 + coeff.z * res2) / 32767.0f;
}

// readFromSDF_color4u_interpolated -> (r,g,b,255)/255
static inline V4f read_colour_interp(const oracle_scene *s, const V3f &pt, IndexCache &c) {
  V3f coeff; V3i pos; bool found;
  floor3(pt, pos, coeff);
  float rx = 0.0f, ry = 0.0f, rz = 0.0f;
  const int off[8][3] = {{0,0,0},{1,0,0},{0,1,0},{1,1,0},{0,0,1},{1,0,1},{0,1,1},{1,1,1}};
  for (int k = 0; k < 8; k++) {
    V3i q = {pos.x + off[k][0], pos.y + off[k][1], pos.z + off[k][2]};
    dslam_voxel v = read_voxel(s, q, found, c);
    float wx = off[k][0] ? coeff.x : (1.0f - coeff.x);
    float wy = off[k][1] ? coeff.y : (1.0f - coeff.y);
    float wz = off[k][2] ? coeff.z : (1.0f - coeff.z);
    float w = wx * wy * wz;
    rx += w * (float)v.clr[0]; ry += w * (float)v.clr[1]; rz += w * (float)v.clr[2];
  }
  V4f r = {rx / 255.0f, ry / 255.0f, rz / 255.0f, 255.0f / 255.0f};
  return r;
}

// computeSingleNormalFromSDF (upstream ITMRepresentationAccess.h), un-normalised gradient
static inline V3f normal_from_sdf(const oracle_scene *s, const V3f &pt, IndexCache &c) {
  bool f;
  V3f ret; V3f coeff; V3i pos;
  floor3(pt, pos, coeff);
  V3f nc = {1.0f - coeff.x, 1.0f - coeff.y, 1.0f - coeff.z};
  V4f front, back, tmp;
  front.x = rd(s, pos, 0, 0, 0, f, c); front.y = rd(s, pos, 1, 0, 0, f, c);
  front.z = rd(s, pos, 0, 1, 0, f, c); front.w = rd(s, pos, 1, 1, 0, f, c);
  back.x = rd(s, pos, 0, 0, 1, f, c); back.y = rd(s, pos, 1, 0, 1, f, c);
  back.z = rd(s, pos, 0, 1, 1, f, c); back.w = rd(s, pos, 1, 1, 1, f, c);
  float p1, p2, v1;
  // gradient x
  p1 = front.x * nc.y * nc.z + front.z * coeff.y * nc.z + back.x * nc.y * coeff.z + back.z * coeff.y * coeff.z;
  tmp.x = rd(s, pos, -1, 0, 0, f, c); tmp.y = rd(s, pos, -1, 1, 0, f, c);
  tmp.z = rd(s, pos, -1, 0, 1, f, c); tmp.w = rd(s, pos, -1, 1, 1, f, c);
  p2 = tmp.x * nc.y * nc.z + tmp.y * coeff.y * nc.z + tmp.z * nc.y * coeff.z + tmp.w * coeff.y * coeff.z;
  v1 = p1 * coeff.x + p2 * nc.x;
  p1 = front.y * nc.y * nc.z + front.w * coeff.y * nc.z + back.y * nc.y * coeff.z + back.w * coeff.y * coeff.z;
  tmp.x = rd(s, pos, 2, 0, 0, f, c); tmp.y = rd(s, pos, 2, 1, 0, f, c);
  tmp.z = rd(s, pos, 2, 0, 1, f, c); tmp.w = rd(s, pos, 2, 1, 1, f, c);
  p2 = tmp.x * nc.y * nc.z + tmp.y * coeff.y * nc.z + tmp.z * nc.y * coeff.z + tmp.w * coeff.y * coeff.z;
  ret.x = (p1 * nc.x + p2 * coeff.x - v1) / 32767.0f;
  // gradient y
  p1 = front.x * nc.x * nc.z + front.y * coeff.x * nc.z + back.x * nc.x * coeff.z + back.y * coeff.x * coeff.z;
  tmp.x = rd(s, pos, 0, -1, 0, f, c); tmp.y = rd(s, pos, 1, -1, 0, f, c);
  tmp.z = rd(s, pos, 0, -1, 1, f, c); tmp.w = rd(s, pos, 1, -1, 1, f, c);
  p2 = tmp.x * nc.x * nc.z + tmp.y * coeff.x * nc.z + tmp.z * nc.x * coeff.z + tmp.w * coeff.x * coeff.z;
  v1 = p1 * coeff.y + p2 * nc.y;
  p1 = front.z * nc.x * nc.z + front.w * coeff.x * nc.z + back.z * nc.x * coeff.z + back.w * coeff.x * coeff.z;
  tmp.x = rd(s, pos, 0, 2, 0, f, c); tmp.y = rd(s, pos, 1, 2, 0, f, c);
  tmp.z = rd(s, pos, 0, 2, 1, f, c); tmp.w = rd(s, pos, 1, 2, 1, f, c);
  p2 = tmp.x * nc.x * nc.z + tmp.y * coeff.x * nc.z + tmp.z * nc.x * coeff.z + tmp.w * coeff.x * coeff.z;
  ret.y = (p1 * nc.y + p2 * coeff.y - v1) / 32767.0f;
  // gradient z
  p1 = front.x * nc.x * nc.y + front.y * coeff.x * nc.y + front.z * nc.x * coeff.y + front.w * coeff.x * coeff.y;
  tmp.x = rd(s, pos, 0, 0, -1, f, c); tmp.y = rd(s, pos, 1, 0, -1, f, c);
  tmp.z = rd(s, pos, 0, 1, -1, f, c); tmp.w = rd(s, pos, 1, 1, -1, f, c);
  p2 = tmp.x * nc.x * nc.y + tmp.y * coeff.x * nc.y + tmp.z * nc.x * coeff.y + tmp.w * coeff.x * coeff.y;
  v1 = p1 * coeff.z + p2 * nc.z;
  p1 = back.x * nc.x * nc.y + back.y * coeff.x * nc.y + back.z * nc.x * coeff.y + back.w * coeff.x * coeff.y;
  tmp.x = rd(s, pos, 0, 0, 2, f, c); tmp.y = rd(s, pos, 1, 0, 2, f, c);
  tmp.z = rd(s, pos, 0, 1, 2, f, c); tmp.w = rd(s, pos, 1, 1, 2, f, c);
  p2 = tmp.x * nc.x * nc.y + tmp.y * coeff.x * nc.y + tmp.z * nc.x * coeff.y + tmp.w * coeff.x * coeff.y;
  ret.z = (p1 * nc.z + p2 * coeff.z - v1) / 32767.0f;
  return ret;
}

// ------------------------------------------------------------------------------------------------
// block visibility
// ------------------------------------------------------------------------------------------------
static inline void check_point_vis(bool &vis, bool &vis_enl, const V4f &pt, const float *M, const float *proj,
                                   int W, int H, bool use_swapping) {
  V4f b = mul(M, pt);
  if (b.z < 1e-10f) return;
  b.x = proj[0] * b.x / b.z + proj[2];
  b.y = proj[1] * b.y / b.z + proj[3];
  if (b.x >= 0 && b.x < W && b.y >= 0 && b.y < H) { vis = true; vis_enl = true; }
  else if (use_swapping) {
    int lx = -W / 8, ly = W + W / 8, lz = -H / 8, lw = H + H / 8;
    if (b.x >= lx && b.x < ly && b.y >= lz && b.y < lw) vis_enl = true;
  }
}

static inline void check_block_vis(bool &vis, bool &vis_enl, const int16_t *pos, const float *M, const float *proj,
                                   float voxel_size, int W, int H, bool use_swapping) {
  V4f pt;
  float factor = (float)DSLAM_BLOCK_SIZE * voxel_size;
  vis = false; vis_enl = false;
  pt.x = (float)pos[0] * factor; pt.y = (float)pos[1] * factor; pt.z = (float)pos[2] * factor; pt.w = 1.0f;
  check_point_vis(vis, vis_enl, pt, M, proj, W, H, use_swapping); if (vis) return;  // 0 0 0
  pt.z += factor; check_point_vis(vis, vis_enl, pt, M, proj, W, H, use_swapping); if (vis) return;  // 0 0 1
  pt.y += factor; check_point_vis(vis, vis_enl, pt, M, proj, W, H, use_swapping); if (vis) return;  // 0 1 1
  pt.x += factor; check_point_vis(vis, vis_enl, pt, M, proj, W, H, use_swapping); if (vis) return;  // 1 1 1
  pt.z -= factor; check_point_vis(vis, vis_enl, pt, M, proj, W, H, use_swapping); if (vis) return;  // 1 1 0
  pt.y -= factor; check_point_vis(vis, vis_enl, pt, M, proj, W, H, use_swapping); if (vis) return;  // 1 0 0
  pt.x -= factor; pt.y += factor; check_point_vis(vis, vis_enl, pt, M, proj, W, H, use_swapping); if (vis) return;  // 0 1 0
  pt.x += factor; pt.y -= factor; pt.z += factor; check_point_vis(vis, vis_enl, pt, M, proj, W, H, use_swapping);  // 1 0 1
}

// ------------------------------------------------------------------------------------------------
// hash-table maintenance shared by decay and sliding window (DESIGN.md "batch removal")
// ------------------------------------------------------------------------------------------------
static const dslam_hash_entry kEmptyEntry = {{0, 0, 0}, 0, 0, -2};

static inline uint64_t *slot_mask(oracle_scene *s, int slot, int ring) {
  return &s->masks[((size_t)slot * 2 + ring) * s->history_words];
}
static inline bool slot_referenced(oracle_scene *s, int slot) {
  const uint64_t *m = slot_mask(s, slot, 0);
  for (int i = 0; i < 2 * s->history_words; i++) if (m[i]) return true;
  return false;
}
static inline void slot_forget(oracle_scene *s, int slot) {
  uint64_t *m = slot_mask(s, slot, 0);
  for (int i = 0; i < 2 * s->history_words; i++) m[i] = 0;
  s->last_seen[slot] = -1;
}
static inline int history_bits(const oracle_scene *s) { return 64 * s->history_words; }

// Release a batch of entries (ascending entry index, all resident).  Voxel-block slots go back to the pool
// in batch order.  Every affected bucket chain is rewritten once: surviving entries keep their chain
// order, the first survivor moves into the bucket head if the head was released, and the excess slots
// that become free are pushed onto the excess free list in ascending slot order.
static void remove_entries(oracle_scene *s, oracle_render_state *r, const std::vector<int> &batch) {
  if (batch.empty()) return;
  const int nb = s->p.num_buckets;
  std::vector<uint8_t> flag(s->n_entries, 0);
  std::vector<int> buckets;
  for (int t : batch) {
    const dslam_hash_entry e = s->hash[t];
    dslam_voxel *vb = &s->vba[(size_t)e.ptr * 512];
    for (int i = 0; i < 512; i++) vb[i] = kEmptyVoxel;
    s->alloc_list[++s->last_free] = e.ptr;
    slot_forget(s, e.ptr);
    flag[t] = 1;
    buckets.push_back(t < nb ? t : hash_index(e.pos[0], e.pos[1], e.pos[2], (uint32_t)(nb - 1)));
  }
  std::sort(buckets.begin(), buckets.end());
  buckets.erase(std::unique(buckets.begin(), buckets.end()), buckets.end());
  std::vector<int> freed;
  for (int head : buckets) {
    int c = head, prev = -1;
    while (c >= 0) {
      const dslam_hash_entry e = s->hash[c];
      const int next = (e.offset >= 1) ? nb + e.offset - 1 : -1;
      if (flag[c]) {
        if (c != head) freed.push_back(c - nb);
        s->hash[c] = kEmptyEntry;
        if (r) r->visible_type[c] = 0;
      } else {
        int cur = c;
        if (prev == -1) {
          if (c != head) {  // first survivor moves into the released bucket head
            s->hash[head] = e;
            if (r) { r->visible_type[head] = r->visible_type[c]; r->visible_type[c] = 0; }
            s->hash[c] = kEmptyEntry;
            freed.push_back(c - nb);
            cur = head;
          }
        } else {
          s->hash[prev].offset = (c - nb) + 1;
        }
        prev = cur;
      }
      c = next;
    }
    if (prev >= 0) s->hash[prev].offset = 0;
  }
  std::sort(freed.begin(), freed.end());
  for (int x : freed) s->excess_list[++s->last_free_ex] = x;
  if (r) {
    int n = 0;
    for (int t = 0; t < r->n_entries; t++) if (r->visible_type[t] > 0) push_visible(r, n, t);
    r->no_visible = std::min(n, r->n_local);
  }
}


}  // namespace

// =================================================================================================
// C entry points (same shapes as include/dslam_fusion.h, prefix oracle_)
// =================================================================================================
extern "C" int oracle_engine_create(oracle_engine **out) {
  oracle_engine *e = new oracle_engine();
  e->wp.depth_weighting = 0; e->wp.max_new_w = 1; e->wp.max_distance = 1.0f;
  e->threads = 1;
  *out = e;
  return 0;
}
extern "C" int oracle_engine_destroy(oracle_engine *e) { delete e; return 0; }
extern "C" int oracle_engine_set_threads(oracle_engine *e, int n) {
  e->threads = n < 1 ? 1 : n;
#ifdef _OPENMP
  omp_set_num_threads(e->threads);
#endif
  return 0;
}
extern "C" int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_num_procs();
#else
  return 1;
#endif
}
extern "C" int oracle_set_fusion_weight_params(oracle_engine *e, const dslam_weight_params *w) {
  // same contract as dslam_set_fusion_weight_params: a voxel weight is one byte
  if (w->depth_weighting && !(w->max_distance > 0 && w->max_new_w >= 1 && w->max_new_w <= 255)) return DSLAM_ERR_INVALID;
  e->wp = *w;
  return 0;
}

// ResetScene (InfiniTamDriver.h:354-360)
extern "C" int oracle_scene_reset(oracle_engine *, oracle_scene *s) {
  for (auto &v : s->vba) v = kEmptyVoxel;
  for (int i = 0; i < s->p.num_local_blocks; i++) s->alloc_list[i] = i;
  s->last_free = s->p.num_local_blocks - 1;
  for (auto &e : s->hash) e = kEmptyEntry;
  for (int i = 0; i < s->p.num_excess; i++) s->excess_list[i] = i;
  s->last_free_ex = s->p.num_excess - 1;
  std::fill(s->alloc_type.begin(), s->alloc_type.end(), 0);
  std::fill(s->last_seen.begin(), s->last_seen.end(), -1);
  std::fill(s->masks.begin(), s->masks.end(), 0);
  for (int q = 0; q < 2; q++) { s->ring_head[q] = 0; s->ring_next[q] = 0; s->decay_cursor[q] = 0; }
  s->frame_counter = 0; s->decayed_blocks = 0; s->slid_blocks = 0;
  s->alloc_failures = 0; s->last_swapped_in = 0; s->last_swapped_out = 0;
  std::fill(s->swap_state.begin(), s->swap_state.end(), 0);
  std::fill(s->has_stored.begin(), s->has_stored.end(), 0);
  return 0;
}

extern "C" int oracle_scene_create(oracle_engine *e, const dslam_scene_params *p, oracle_scene **out) {
  oracle_scene *s = new oracle_scene();
  s->p = *p;
  if (s->p.num_local_blocks <= 0) s->p.num_local_blocks = DSLAM_DEFAULT_LOCAL_BLOCK_NUM;
  if (s->p.num_buckets <= 0) s->p.num_buckets = DSLAM_DEFAULT_BUCKET_NUM;
  if (s->p.num_excess <= 0) s->p.num_excess = DSLAM_DEFAULT_EXCESS_LIST_SIZE;
  if (s->p.num_buckets & (s->p.num_buckets - 1)) { delete s; return DSLAM_ERR_INVALID; }
  s->n_entries = s->p.num_buckets + s->p.num_excess;
  s->hash.resize(s->n_entries);
  s->vba.resize((size_t)s->p.num_local_blocks * 512);
  s->alloc_list.resize(s->p.num_local_blocks);
  s->excess_list.resize(s->p.num_excess);
  s->alloc_type.resize(s->n_entries);
  s->block_coords.resize(s->n_entries);
  s->last_seen.resize(s->p.num_local_blocks);
  if (s->p.history_words <= 0) s->p.history_words = 4;
  s->history_words = s->p.history_words;
  s->masks.resize((size_t)s->p.num_local_blocks * 2 * s->history_words);
  s->stored = nullptr;
  if (s->p.use_swapping) {
    s->swap_state.resize(s->n_entries);
    s->has_stored.resize(s->n_entries);
    s->stored = (dslam_voxel *)calloc((size_t)s->n_entries * 512, sizeof(dslam_voxel));
    if (!s->stored) { delete s; return DSLAM_ERR_INVALID; }
  }
  s->shard = 0; s->num_shards = 1; s->chunk_blocks = 256;
  s->shard_first = 0; s->shard_count = -1;
  oracle_scene_reset(e, s);
  *out = s;
  return 0;
}
extern "C" int oracle_scene_destroy(oracle_scene *s) { if (s) { free(s->stored); delete s; } return 0; }
extern "C" int oracle_scene_set_shard(oracle_scene *s, int shard, int num_shards, int chunk_blocks) {
  if (s->p.use_swapping && num_shards > 1) return DSLAM_ERR_INVALID;  // (as the engine: the host store is per rank)
  s->shard = shard; s->num_shards = num_shards; s->chunk_blocks = chunk_blocks; return 0;
}

extern "C" int oracle_scene_set_shard_range(oracle_scene *s, int first, int count) {
  if (s->p.use_swapping && count >= 0) return DSLAM_ERR_INVALID;
  s->shard_first = first; s->shard_count = count; return 0;
}

// ---- the exchange step of the sharded re-integration (same contract as dslam_scene_track_dirty / dslam_shard_dirty_*;
// the "device" buffers are host memory here) --------------------------------------------------------------------------
extern "C" int oracle_scene_track_dirty(oracle_engine *, oracle_scene *s, int enable) {
  if (enable) s->dirty.assign((size_t)s->p.num_local_blocks, 0);
  s->dirty_tracking = enable != 0;
  return 0;
}
extern "C" int oracle_shard_dirty_plan(oracle_engine *, oracle_scene *s, int num_shards, int chunk_blocks, int32_t *counts_out) {
  const int N = s->p.num_local_blocks;
  if (s->dirty.empty() || num_shards < 1 || num_shards > 64 || chunk_blocks < 1 || N % (num_shards * chunk_blocks)) return DSLAM_ERR_INVALID;
  s->dirty_list.clear();
  s->dirty_counts.assign(num_shards, 0);
  for (int r = 0; r < num_shards; r++)
    for (int c = r; c < N / chunk_blocks; c += num_shards)
      for (int slot = c * chunk_blocks; slot < (c + 1) * chunk_blocks; slot++)
        if (s->dirty[slot]) { s->dirty_list.push_back(slot); s->dirty_counts[r]++; }
  for (int r = 0; r < num_shards; r++) counts_out[r] = s->dirty_counts[r];
  s->dirty_chunk = chunk_blocks;
  return 0;
}
extern "C" int oracle_shard_dirty_pack(oracle_engine *, const oracle_scene *s, int shard, void *send, int capacity_blocks) {
  if (s->dirty_counts.empty() || shard < 0 || shard >= (int)s->dirty_counts.size()) return DSLAM_ERR_INVALID;
  size_t off = 0;
  for (int r = 0; r < shard; r++) off += s->dirty_counts[r];
  const int n = std::min(s->dirty_counts[shard], capacity_blocks);
  for (int i = 0; i < n; i++)
    memcpy((char *)send + (size_t)i * 4096, &s->vba[(size_t)s->dirty_list[off + i] * 512], 4096);
  return 0;
}
extern "C" int oracle_shard_dirty_unpack(oracle_engine *, oracle_scene *s, int skip_shard, const void *recv, int stride_blocks) {
  if (s->dirty_counts.empty()) return DSLAM_ERR_INVALID;
  size_t off = 0;
  for (int r = 0; r < (int)s->dirty_counts.size(); r++) {
    const int n = std::min(s->dirty_counts[r], stride_blocks);
    if (r != skip_shard)
      for (int i = 0; i < n; i++)
        memcpy(&s->vba[(size_t)s->dirty_list[off + i] * 512], (const char *)recv + ((size_t)r * stride_blocks + i) * 4096, 4096);
    off += s->dirty_counts[r];
  }
  return 0;
}

extern "C" int oracle_render_state_create(oracle_engine *, const oracle_scene *s, int w, int h, oracle_render_state **out) {
  oracle_render_state *r = new oracle_render_state();
  r->w = w; r->h = h; r->n_entries = s->n_entries; r->n_local = s->p.num_local_blocks;
  r->visible_ids.assign(s->p.num_local_blocks, 0);
  r->no_visible = 0;
  r->visible_type.assign(s->n_entries, 0);
  r->range.assign((size_t)w * h, V2f{0, 0});
  r->raycast.assign((size_t)w * h, V4f{0, 0, 0, 0});
  r->image_rgba.assign((size_t)w * h * 4, 0);
  r->image_float.assign((size_t)w * h, 0.0f);
  *out = r;
  return 0;
}
extern "C" int oracle_render_state_destroy(oracle_render_state *r) { delete r; return 0; }

extern "C" int oracle_view_create(oracle_engine *, int w_rgb, int h_rgb, int w_d, int h_d, oracle_view **out) {
  oracle_view *v = new oracle_view();
  v->w_rgb = w_rgb; v->h_rgb = h_rgb; v->w_d = w_d; v->h_d = h_d;
  v->rgba.assign((size_t)w_rgb * h_rgb * 4, 0);
  v->depth.assign((size_t)w_d * h_d, -1.0f);
  v->timestamp = 0;
  *out = v;
  return 0;
}
extern "C" int oracle_view_destroy(oracle_view *v) { delete v; return 0; }

// exp(x) for x <= 0 as a fixed operation sequence (Cody-Waite reduction, degree-6 polynomial in explicit FMAs, exact
// power-of-two scaling), within 1 ulp of libm's expf (tests/test_oracle_kat.py).  The HIP filter kernel runs the
// same sequence, which makes the bilateral filter -- the only transcendental on the path -- bit-identical on host
// and device.  x < -86 returns 0: next to the centre tap's weight of exactly 1 such a weight never reaches the sums.
static float det_exp(float x) {
  if (x < -86.0f) return 0.0f;
  const float n = rintf(x * 1.44269504f);
  float r = fmaf(-n, 0.693359375f, x);
  r = fmaf(-n, -2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  const float y = fmaf(p, r * r, r) + 1.0f;
  const int32_t bits = ((int32_t)n + 127) << 23;
  float scale;
  memcpy(&scale, &bits, 4);
  return y * scale;
}
extern "C" float oracle_det_exp(float x) { return det_exp(x); }

// filterDepth (upstream InfiniTAM v2 ITMViewBuilder_Shared.h; [UPSTREAM-RECALL]) over the interior; border pixels
// of `out` are left alone, as upstream's loop bounds do
static void filter_depth_pass(const float *in, float *out, int W, int H) {
  const float MEAN_SIGMA_L = 1.2232f;
#pragma omp parallel for schedule(static)
  for (int y = 2; y < H - 2; y++)
    for (int x = 2; x < W - 2; x++) {
      const float z = in[x + y * W];
      if (z < 0.0f) { out[x + y * W] = -1.0f; continue; }
      const float sigma_z = 1.0f / (0.0012f + 0.0019f * (z - 0.4f) * (z - 0.4f) + 0.0001f / sqrtf(z) * 0.25f);
      float final_depth = 0.0f, w_sum = 0.0f;
      for (int i = -2; i <= 2; i++)
        for (int j = -2; j <= 2; j++) {
          const float tmpz = in[(x + j) + (y + i) * W];
          if (tmpz < 0.0f) continue;
          float dz = tmpz - z;
          dz *= dz;
          const float w = det_exp(-0.5f * ((float)(abs(i) + abs(j)) * MEAN_SIGMA_L * MEAN_SIGMA_L + dz * sigma_z * sigma_z));
          w_sum += w;
          final_depth += w * tmpz;
        }
      out[x + y * W] = final_depth / w_sum;
    }
}

static int view_finish(oracle_view *v, const int16_t *depth_mm, float a, float b, double timestamp, int use_bilateral) {
  const size_t n = (size_t)v->w_d * v->h_d;
  v->raw_depth.assign(depth_mm, depth_mm + n);
  for (size_t i = 0; i < n; i++) {
    int d = depth_mm[i];
    v->depth[i] = (d <= 0 || d > 32000) ? -1.0f : (float)d * a + b;
  }
  if (use_bilateral) {
    // ITMViewBuilder::UpdateView: DepthFiltering(floatImage <- depth), (depth <- floatImage), ... five passes, then
    // depth = floatImage.  floatImage is zero-initialised and its border is never written
    if (v->w_d < 5 || v->h_d < 5) return DSLAM_ERR_INVALID;
    if (v->float_image.size() != n) v->float_image.assign(n, 0.0f);
    float *A = v->depth.data(), *B = v->float_image.data();
    filter_depth_pass(A, B, v->w_d, v->h_d);
    filter_depth_pass(B, A, v->w_d, v->h_d);
    filter_depth_pass(A, B, v->w_d, v->h_d);
    filter_depth_pass(B, A, v->w_d, v->h_d);
    filter_depth_pass(A, B, v->w_d, v->h_d);
    v->depth = v->float_image;
  }
  v->timestamp = timestamp;
  return 0;
}

// viewBuilder->UpdateView (InfiniTamDriver.cpp:280-288): convertDepthAffineToFloat
extern "C" int oracle_view_update(oracle_engine *, oracle_view *v, const uint8_t *rgba, const int16_t *depth_mm, float a,
                       float b, double timestamp, int use_bilateral) {
  memcpy(v->rgba.data(), rgba, v->rgba.size());
  return view_finish(v, depth_mm, a, b, timestamp, use_bilateral);
}

// CvToItm(cv::Mat3b) + UpdateView (InfiniTamDriver.cpp:84-103, 280-288)
extern "C" int oracle_view_update_bgr(oracle_engine *, oracle_view *v, const uint8_t *bgr, const int16_t *depth_mm, float a,
                           float b, double timestamp, int use_bilateral) {
  const size_t n = (size_t)v->w_rgb * v->h_rgb;
  for (size_t i = 0; i < n; i++) {
    v->rgba[4 * i + 2] = bgr[3 * i];      // .b = col[0]
    v->rgba[4 * i + 1] = bgr[3 * i + 1];  // .g = col[1]
    v->rgba[4 * i + 0] = bgr[3 * i + 2];  // .r = col[2]
    v->rgba[4 * i + 3] = 255u;
  }
  return view_finish(v, depth_mm, a, b, timestamp, use_bilateral);
}

// PrecomputedDepthProvider::ReadPrecomputed's per-pixel loop on the int16 image (PrecomputedDepthProvider.cpp:30-64)
static int16_t wrap_i16(float f) { return (int16_t)(int32_t)f; }  // the x86 behaviour of the (undefined) overflow
extern "C" int oracle_view_update_dataset(oracle_engine *e, oracle_view *v, const uint8_t *colour, int channels,
                               const int16_t *raw, int format, float max_depth_m, float a, float b, double timestamp,
                               int use_bilateral) {
  if ((channels != 3 && channels != 4) || format < DSLAM_DEPTH_MM || format > DSLAM_DEPTH_RGBD_X5) return DSLAM_ERR_INVALID;
  const size_t n = (size_t)v->w_d * v->h_d;
  std::vector<int16_t> mm(raw, raw + n);
  const float kitti_factor = 1000.0 / 256.0;
  const float max_depth_mm_f = max_depth_m * 1000.0f;
  const int16_t max_depth_mm_s = static_cast<int16_t>(round(max_depth_mm_f));
  if (format == DSLAM_DEPTH_KITTI_X256) {
    for (size_t i = 0; i < n; i++) {
      if (mm[i] > max_depth_m * 256) mm[i] = 0;
      mm[i] = wrap_i16((float)mm[i] * kitti_factor);
    }
  } else if (format == DSLAM_DEPTH_RGBD_X5) {
    for (size_t i = 0; i < n; i++) {
      mm[i] = static_cast<int16_t>(((float)mm[i]) / 5.0);
      if (mm[i] > max_depth_mm_s) mm[i] = 0;
    }
  }
  if (channels == 3) return oracle_view_update_bgr(e, v, colour, mm.data(), a, b, timestamp, use_bilateral);
  return oracle_view_update(e, v, colour, mm.data(), a, b, timestamp, use_bilateral);
}
extern "C" int oracle_download_view_raw_depth(oracle_engine *, const oracle_view *v, int16_t *out) {
  memcpy(out, v->raw_depth.data(), v->raw_depth.size() * 2);
  return 0;
}

extern "C" int oracle_download_view_rgba(oracle_engine *, const oracle_view *v, uint8_t *out) {
  memcpy(out, v->rgba.data(), v->rgba.size());
  return 0;
}

// keyframe store (the images of DenseSlam's mfusionFrameDataBase, DenseSlam.h:46-60,431-433): plain host vectors here
struct oracle_frame_store {
  int w_rgb, h_rgb, w_d, h_d, capacity;
  std::vector<std::vector<uint8_t>> rgba;
  std::vector<std::vector<int16_t>> depth;
  // optional: the visible list of each keyframe's fusion, with the block position every entry held
  bool lists_enabled = false;
  std::vector<std::vector<int32_t>> list_ids;
  std::vector<std::vector<S4>> list_pos;
  std::vector<uint8_t> has_list;
};
extern "C" int oracle_frame_store_create(oracle_engine *, int w_rgb, int h_rgb, int w_d, int h_d, int capacity, oracle_frame_store **out) {
  if (capacity <= 0 || w_rgb <= 0 || h_rgb <= 0 || w_d <= 0 || h_d <= 0) return DSLAM_ERR_INVALID;
  oracle_frame_store *fs = new oracle_frame_store();
  fs->w_rgb = w_rgb; fs->h_rgb = h_rgb; fs->w_d = w_d; fs->h_d = h_d; fs->capacity = capacity;
  fs->rgba.assign(capacity, std::vector<uint8_t>((size_t)w_rgb * h_rgb * 4, 0));
  fs->depth.assign(capacity, std::vector<int16_t>((size_t)w_d * h_d, 0));
  *out = fs;
  return 0;
}
extern "C" int oracle_frame_store_destroy(oracle_frame_store *fs) { delete fs; return 0; }
extern "C" int oracle_frame_store_put(oracle_engine *, oracle_frame_store *fs, int slot, const uint8_t *rgba, const int16_t *depth) {
  if (slot < 0 || slot >= fs->capacity) return DSLAM_ERR_INVALID;
  memcpy(fs->rgba[slot].data(), rgba, fs->rgba[slot].size());
  memcpy(fs->depth[slot].data(), depth, fs->depth[slot].size() * 2);
  return 0;
}
extern "C" int oracle_frame_store_put_bgr(oracle_engine *, oracle_frame_store *fs, int slot, const uint8_t *bgr, const int16_t *depth) {
  if (slot < 0 || slot >= fs->capacity) return DSLAM_ERR_INVALID;
  const size_t n = (size_t)fs->w_rgb * fs->h_rgb;
  uint8_t *o = fs->rgba[slot].data();
  for (size_t i = 0; i < n; i++) { o[4 * i] = bgr[3 * i + 2]; o[4 * i + 1] = bgr[3 * i + 1]; o[4 * i + 2] = bgr[3 * i]; o[4 * i + 3] = 255u; }
  memcpy(fs->depth[slot].data(), depth, fs->depth[slot].size() * 2);
  return 0;
}
extern "C" int oracle_frame_store_put_view(oracle_engine *, oracle_frame_store *fs, int slot, const oracle_view *v) {
  if (slot < 0 || slot >= fs->capacity) return DSLAM_ERR_INVALID;
  if (v->w_rgb != fs->w_rgb || v->h_rgb != fs->h_rgb || v->w_d != fs->w_d || v->h_d != fs->h_d) return DSLAM_ERR_INVALID;
  if (v->raw_depth.size() != fs->depth[slot].size()) return DSLAM_ERR_INVALID;  // view never updated
  fs->rgba[slot] = v->rgba;
  fs->depth[slot] = v->raw_depth;
  return 0;
}
extern "C" int oracle_frame_store_get(oracle_engine *, const oracle_frame_store *fs, int slot, uint8_t *rgba_out, int16_t *depth_out) {
  if (slot < 0 || slot >= fs->capacity) return DSLAM_ERR_INVALID;
  if (rgba_out) memcpy(rgba_out, fs->rgba[slot].data(), fs->rgba[slot].size());
  if (depth_out) memcpy(depth_out, fs->depth[slot].data(), fs->depth[slot].size() * 2);
  return 0;
}
extern "C" int oracle_view_update_from_store(oracle_engine *, oracle_view *v, const oracle_frame_store *fs, int slot, float a, float b,
                                  double timestamp, int use_bilateral) {
  if (slot < 0 || slot >= fs->capacity) return DSLAM_ERR_INVALID;
  if (v->w_rgb != fs->w_rgb || v->h_rgb != fs->h_rgb || v->w_d != fs->w_d || v->h_d != fs->h_d) return DSLAM_ERR_INVALID;
  memcpy(v->rgba.data(), fs->rgba[slot].data(), v->rgba.size());
  return view_finish(v, fs->depth[slot].data(), a, b, timestamp, use_bilateral);
}

// DenseSlam::depthPostProcessing, the pixel loop (DenseSlam.cpp:488-529).  Arithmetic types follow the C++ of the
// reference: floats from cv::Mat CV_32F elements, doubles where a double literal enters the expression; cv::Mat
// products of CV_32F operands accumulate in double and round once (OpenCV's GEMMSingleMul<float,double>) [recalled:
// OpenCV is not in this image].  `row` goes with cx/fx and `col` with cy/fy, as written there.  The double -> int
// conversions saturate (undefined in C when out of range; such projections fail the bounds test either way).
static int d2i_sat(double v) {
  if (!(v == v)) return INT_MIN;
  if (v >= 2147483647.0) return INT_MAX;
  if (v <= -2147483648.0) return INT_MIN;
  return (int)v;
}
extern "C" int oracle_depth_post_processing(oracle_engine *, int16_t *curr, const int16_t *prev_s, int cols, int rows,
                                 const float *Tpc, const float *intr, float threshold, float area, int *count_out) {
  const uint16_t *prev = (const uint16_t *)prev_s;  // prev_depth.at<uint16_t>(row_u, col_v)  (:515)
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const float inv_fx = (float)(1.0 / fx), inv_fy = (float)(1.0 / fy);  // (:440-441)
  float R[9], t[3];
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) R[3 * r + c] = Tpc[c * 4 + r];
    t[r] = Tpc[12 + r];
  }
  int count = 0;
  for (int row = 0; row < rows; row++)
    for (int col = 0; col < cols; col++) {
      const float z = (float)(((float)curr[row * cols + col]) / 1000.0);
      if (z < 0.005) continue;
      const float X = z * (row - cx) * inv_fx;
      const float Y = z * (col - cy) * inv_fy;
      float P[3];
      for (int k = 0; k < 3; k++) {
        const double acc = (double)R[3 * k] * X + (double)R[3 * k + 1] * Y + (double)R[3 * k + 2] * z;
        P[k] = (float)acc + t[k];
      }
      const int row_u = d2i_sat(fx * P[0] * (1.0 / P[2]) + cx + 0.5);
      const int col_v = d2i_sat(fy * P[1] * (1.0 / P[2]) + cy + 0.5);
      if (row_u < 0.1 || col_v < 0.1 || (row_u + 1) > rows || (col_v + 1) > cols) continue;
      const float prev_z = (float)((float)prev[row_u * cols + col_v] / 1000.0);
      if (prev_z < 0.005) continue;
      const float curr_z = P[2];
      const float diff = fabsf(prev_z - curr_z);
      if ((diff / curr_z) > threshold && row > area * rows) curr[row * cols + col] = 0;
      count++;
    }
  if (count_out) *count_out = count;
  return 0;
}

// -------------------------------------------------------------------------------------------------
// AllocateSceneFromDepth (reached from denseMapper->ProcessFrame, InfiniTamDriver.h:187-192)
// -------------------------------------------------------------------------------------------------
extern "C" int oracle_allocate_scene_from_depth(oracle_engine *, oracle_scene *s, const oracle_view *v, oracle_render_state *r,
                                     const float *M_d, const float *intr, int only_update_visible_list) {
  const int W = v->w_d, H = v->h_d;
  const float mu = s->p.mu, vs = s->p.voxel_size;
  const int nb = s->p.num_buckets;
  const uint32_t mask = (uint32_t)(nb - 1);
  const bool use_swapping = s->p.use_swapping != 0;
  float invM[16];
  inv4(M_d, invM);
  const float inv_fx = 1.0f / intr[0], inv_fy = 1.0f / intr[1], cx = intr[2], cy = intr[3];
  const float one_over_block = 1.0f / (vs * DSLAM_BLOCK_SIZE);

  std::fill(s->alloc_type.begin(), s->alloc_type.end(), 0);
  for (int i = 0; i < r->no_visible; i++) r->visible_type[r->visible_ids[i]] = 3;

  // MARK: buildHashAllocAndVisibleTypePP for every pixel, sequential row-major (last writer wins)
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      float d = v->depth[x + y * W];
      if (d <= 0 || (d - mu) < 0 || (d - mu) < s->p.frustum_min || (d + mu) > s->p.frustum_max) continue;
      V3f pc;
      pc.z = d;
      pc.x = pc.z * (((float)x - cx) * inv_fx);
      pc.y = pc.z * (((float)y - cy) * inv_fy);
      float norm = sqrtf(pc.x * pc.x + pc.y * pc.y + pc.z * pc.z);
      V4f tmp;
      tmp.x = pc.x * (1.0f - mu / norm); tmp.y = pc.y * (1.0f - mu / norm); tmp.z = pc.z * (1.0f - mu / norm); tmp.w = 1.0f;
      V4f q = mul(invM, tmp);
      V3f pt = {q.x * one_over_block, q.y * one_over_block, q.z * one_over_block};
      tmp.x = pc.x * (1.0f + mu / norm); tmp.y = pc.y * (1.0f + mu / norm); tmp.z = pc.z * (1.0f + mu / norm);
      q = mul(invM, tmp);
      V3f pe = {q.x * one_over_block, q.y * one_over_block, q.z * one_over_block};
      V3f dir = {pe.x - pt.x, pe.y - pt.y, pe.z - pt.z};
      norm = sqrtf(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
      int no_steps = (int)ceilf(2.0f * norm);
      float div = (float)(no_steps - 1);
      dir.x /= div; dir.y /= div; dir.z /= div;
      for (int i = 0; i < no_steps; i++) {
        int16_t bx = (int16_t)floorf(pt.x), by = (int16_t)floorf(pt.y), bz = (int16_t)floorf(pt.z);
        int h = hash_index(bx, by, bz, mask);
        dslam_hash_entry e = s->hash[h];
        bool found = false;
        if (e.pos[0] == bx && e.pos[1] == by && e.pos[2] == bz && e.ptr >= -1) {
          r->visible_type[h] = (e.ptr == -1) ? 2 : 1;
          found = true;
        }
        if (!found) {
          bool excess = false;
          if (e.ptr >= -1) {
            while (e.offset >= 1) {
              h = nb + e.offset - 1;
              e = s->hash[h];
              if (e.pos[0] == bx && e.pos[1] == by && e.pos[2] == bz && e.ptr >= -1) {
                r->visible_type[h] = (e.ptr == -1) ? 2 : 1;
                found = true;
                break;
              }
            }
            excess = true;
          }
          if (!found) {
            s->alloc_type[h] = excess ? 2 : 1;
            if (!excess) r->visible_type[h] = 1;
            s->block_coords[h] = S4{bx, by, bz, 1};
          }
        }
        pt.x += dir.x; pt.y += dir.y; pt.z += dir.z;
      }
    }

  // COMMIT: allocateVoxelBlocksList, ascending entry index
  s->alloc_failures = 0;
  if (!only_update_visible_list) {
    for (int t = 0; t < s->n_entries; t++) {
      int vba_idx, ex_idx;
      switch (s->alloc_type[t]) {
        case 1:
          vba_idx = s->last_free; s->last_free--;
          if (vba_idx >= 0) {
            dslam_hash_entry he;
            he.pos[0] = s->block_coords[t].x; he.pos[1] = s->block_coords[t].y; he.pos[2] = s->block_coords[t].z;
            he._pad = 0; he.ptr = s->alloc_list[vba_idx]; he.offset = 0;
            s->hash[t] = he;
          } else {
            r->visible_type[t] = 0;
            s->last_free++;
            s->alloc_failures++;
          }
          break;
        case 2:
          vba_idx = s->last_free; s->last_free--;
          ex_idx = s->last_free_ex; s->last_free_ex--;
          if (vba_idx >= 0 && ex_idx >= 0) {
            dslam_hash_entry he;
            he.pos[0] = s->block_coords[t].x; he.pos[1] = s->block_coords[t].y; he.pos[2] = s->block_coords[t].z;
            he._pad = 0; he.ptr = s->alloc_list[vba_idx]; he.offset = 0;
            int ex_off = s->excess_list[ex_idx];
            s->hash[t].offset = ex_off + 1;
            s->hash[nb + ex_off] = he;
            r->visible_type[nb + ex_off] = 1;
          } else {
            s->last_free++; s->last_free_ex++;
            s->alloc_failures++;
          }
          break;
        default: break;
      }
    }
  }

  // VISIBLE LIST: buildVisibleList, ascending entry index
  int n = 0;
  for (int t = 0; t < s->n_entries; t++) {
    uint8_t vt = r->visible_type[t];
    const dslam_hash_entry &e = s->hash[t];
    if (vt == 3) {
      bool vis, vis_enl;
      check_block_vis(vis, vis_enl, e.pos, M_d, intr, vs, W, H, use_swapping);
      if (use_swapping) { if (!vis_enl) vt = 0; } else { if (!vis) vt = 0; }
      r->visible_type[t] = vt;
    }
    if (use_swapping) { if (vt > 0 && s->swap_state[t] != 2) s->swap_state[t] = 1; }
    if (vt > 0) push_visible(r, n, t);
  }
  r->no_visible = std::min(n, r->n_local);

  // REALLOC swapped-out blocks that came back into view
  if (use_swapping) {
    for (int t = 0; t < s->n_entries; t++) {
      if (r->visible_type[t] > 0 && s->hash[t].ptr == -1) {
        int vba_idx = s->last_free; s->last_free--;
        if (vba_idx >= 0) s->hash[t].ptr = s->alloc_list[vba_idx];
        else s->last_free++;
      }
    }
  }
  return 0;
}

// -------------------------------------------------------------------------------------------------
// IntegrateIntoScene / de-integration
// -------------------------------------------------------------------------------------------------
namespace {

static inline void bilinear_rgb(const uint8_t *rgba, float px, float py, int W, float out[3]) {
  int ix = (int)floorf(px), iy = (int)floorf(py);
  float dx = px - (float)ix, dy = py - (float)iy;
  const uint8_t *a = rgba + 4 * ((size_t)ix + (size_t)iy * W);
  const uint8_t *b = rgba + 4 * ((size_t)(ix + 1) + (size_t)iy * W);
  const uint8_t *c = rgba + 4 * ((size_t)ix + (size_t)(iy + 1) * W);
  const uint8_t *d = rgba + 4 * ((size_t)(ix + 1) + (size_t)(iy + 1) * W);
  for (int k = 0; k < 3; k++)
    out[k] = ((float)a[k] * (1.0f - dx) * (1.0f - dy) + (float)b[k] * dx * (1.0f - dy) + (float)c[k] * (1.0f - dx) * dy +
              (float)d[k] * dx * dy);
}

// per-measurement weight (SURVEY A.11 WeightParams law; the fork's law is not knowable)
static inline int new_weight(const dslam_weight_params &wp, float depth_measure) {
  if (!wp.depth_weighting) return 1;
  float dd = depth_measure < wp.max_distance ? depth_measure : wp.max_distance;
  int w = (int)roundf((float)wp.max_new_w * (1.0f - dd / wp.max_distance));
  return w < 1 ? 1 : (w > wp.max_new_w ? wp.max_new_w : w);  // (the upper clamp never acts for a measured depth > 0)
}

// Upstream skips a voxel with `pt_camera.z <= 0` and then tests `u < 1 || u > W - 2 || ...`, which NaN passes.  Here a
// voxel is skipped unless its camera depth is a NORMAL positive float (a denormal z -- the voxel centre in the camera
// plane to within 1e-38 m -- is "not in front of the camera"), and the image-bounds tests are written so that NaN
// fails them: no pose, however degenerate, can index the depth / colour image with (int)NaN.  Identical to upstream
// for every finite, non-degenerate input.  The HIP kernel states the same two rules (csrc/integrate.hip kMinCamZ).
static constexpr float kMinCamZ = 1.17549435e-38f;  // FLT_MIN
static inline bool in_image(float u, float w, float umax, float wmax) { return u >= 1.0f && u <= umax && w >= 1.0f && w <= wmax; }

template <bool DEINTEGRATE>
static inline void update_voxel(dslam_voxel &vox, const V4f &pt_model, const float *M_d, const float *proj_d,
                                const float *M_rgb, const float *proj_rgb, float mu, int maxW, const float *depth,
                                int Wd, int Hd, const uint8_t *rgba, int Wr, int Hr, const dslam_weight_params &wp) {
  // computeUpdatedVoxelDepthInfo
  float eta;
  {
    V4f pc = mul(M_d, pt_model);
    if (!(pc.z >= kMinCamZ)) return;
    float u = proj_d[0] * pc.x / pc.z + proj_d[2];
    float w = proj_d[1] * pc.y / pc.z + proj_d[3];
    if (!in_image(u, w, (float)(Wd - 2), (float)(Hd - 2))) return;
    float dm = depth[(int)(u + 0.5f) + (int)(w + 0.5f) * Wd];
    if (dm <= 0.0f) return;
    eta = dm - pc.z;
    if (eta < -mu) return;
    float oldF = sdf_to_float(vox.sdf);
    int oldW = vox.w_depth;
    float newF = std::min(1.0f, eta / mu);
    int newW = new_weight(wp, dm);
    if (!DEINTEGRATE) {
      newF = (float)oldW * oldF + (float)newW * newF;
      newW = oldW + newW;
      newF /= (float)newW;
      newW = std::min(newW, maxW);
      vox.sdf = float_to_sdf(newF);
      vox.w_depth = (uint8_t)newW;
    } else {
      if (oldW >= newW) {
        int remW = oldW - newW;
        if (remW == 0) { vox.sdf = 32767; vox.w_depth = 0; }
        else {
          float F = ((float)oldW * oldF - (float)newW * newF) / (float)remW;
          F = std::max(-1.0f, std::min(1.0f, F));
          vox.sdf = float_to_sdf(F);
          vox.w_depth = (uint8_t)remW;
        }
      }
    }
  }
  if ((eta > mu) || (fabsf(eta / mu) > 0.25f)) return;
  // computeUpdatedVoxelColorInfo
  {
    V4f pc = mul(M_rgb, pt_model);
    float u = proj_rgb[0] * pc.x / pc.z + proj_rgb[2];
    float w = proj_rgb[1] * pc.y / pc.z + proj_rgb[3];
    if (!in_image(u, w, (float)(Wr - 2), (float)(Hr - 2))) return;
    float m[3];
    bilinear_rgb(rgba, u, w, Wr, m);
    float oldW = (float)vox.w_color;
    if (!DEINTEGRATE) {
      float newW = oldW + 1.0f;
      for (int k = 0; k < 3; k++) {
        float oldC = (float)vox.clr[k] / 255.0f;
        float c = m[k] / 255.0f;
        float nc = (oldC * oldW + c * 1.0f) / newW;
        vox.clr[k] = (uint8_t)(nc * 255.0f);
      }
      newW = std::min(newW, (float)maxW);
      vox.w_color = (uint8_t)newW;
    } else {
      if (vox.w_color >= 1) {
        float remW = oldW - 1.0f;
        if (remW == 0.0f) { vox.clr[0] = vox.clr[1] = vox.clr[2] = 0; vox.w_color = 0; }
        else {
          for (int k = 0; k < 3; k++) {
            float oldC = (float)vox.clr[k] / 255.0f;
            float c = m[k] / 255.0f;
            float nc = (oldC * oldW - c * 1.0f) / remW;
            nc = std::max(0.0f, std::min(1.0f, nc));
            vox.clr[k] = (uint8_t)(nc * 255.0f);
          }
          vox.w_color = (uint8_t)remW;
        }
      }
    }
  }
}

template <bool DEINTEGRATE>
static void integrate_impl(oracle_engine *e, oracle_scene *s, const oracle_view *v, const oracle_render_state *r,
                           const float *M_d, const float *intr_d, const float *M_rgb_in, const float *intr_rgb_in) {
  const float *M_rgb = M_rgb_in ? M_rgb_in : M_d;
  const float *intr_rgb = intr_rgb_in ? intr_rgb_in : intr_d;
  const float vs = s->p.voxel_size, mu = s->p.mu;
  const int maxW = s->p.max_w;
  const bool stop_max = s->p.stop_integrating_at_max_w != 0;
  const int n = r->no_visible;
#pragma omp parallel for schedule(dynamic, 64) if (e->threads > 1)
  for (int i = 0; i < n; i++) {
    const dslam_hash_entry &he = s->hash[r->visible_ids[i]];
    if (he.ptr < 0) continue;
    if (s->dirty_tracking) s->dirty[he.ptr] = 1;  // (before the shard test: every rank ends up with the same marks)
    if (s->num_shards > 1 && ((he.ptr / s->chunk_blocks) % s->num_shards) != s->shard) continue;
    if (s->shard_count >= 0 && (he.ptr < s->shard_first || he.ptr >= s->shard_first + s->shard_count)) continue;
    int gx = he.pos[0] * DSLAM_BLOCK_SIZE, gy = he.pos[1] * DSLAM_BLOCK_SIZE, gz = he.pos[2] * DSLAM_BLOCK_SIZE;
    dslam_voxel *vb = &s->vba[(size_t)he.ptr * 512];
    for (int z = 0; z < 8; z++)
      for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) {
          int loc = x + y * 8 + z * 64;
          if (!DEINTEGRATE && stop_max && vb[loc].w_depth == maxW) continue;
          V4f pm = {(float)(gx + x) * vs, (float)(gy + y) * vs, (float)(gz + z) * vs, 1.0f};
          update_voxel<DEINTEGRATE>(vb[loc], pm, M_d, intr_d, M_rgb, intr_rgb, mu, maxW, v->depth.data(), v->w_d, v->h_d,
                                    v->rgba.data(), v->w_rgb, v->h_rgb, e->wp);
        }
  }
}

// queue the frame's visible list (SURVEY A.9 / A.11 isDefusion routing): set this list's bit on every
// resident visible block.  A full ring drops its oldest list without releasing anything.
static void push_visible_list(oracle_scene *s, const oracle_render_state *r, int q) {
  const int bits = history_bits(s);
  if (s->ring_next[q] - s->ring_head[q] == bits) {
    const int k = s->ring_head[q] % bits;
    for (int slot = 0; slot < s->p.num_local_blocks; slot++) slot_mask(s, slot, q)[k >> 6] &= ~(1ull << (k & 63));
    s->ring_head[q]++;
    if (s->decay_cursor[q] < s->ring_head[q]) s->decay_cursor[q] = s->ring_head[q];
  }
  const int k = (s->ring_next[q]++) % bits;
  const int frame = s->frame_counter++;
  for (int i = 0; i < r->no_visible; i++) {
    const dslam_hash_entry &e = s->hash[r->visible_ids[i]];
    if (e.ptr < 0) continue;
    slot_mask(s, e.ptr, q)[k >> 6] |= (1ull << (k & 63));
    s->last_seen[e.ptr] = frame;
  }
}

}  // namespace

extern "C" int oracle_integrate_into_scene(oracle_engine *e, oracle_scene *s, const oracle_view *v, const oracle_render_state *r,
                                const float *M_d, const float *intr_d, const float *M_rgb, const float *intr_rgb) {
  integrate_impl<false>(e, s, v, r, M_d, intr_d, M_rgb, intr_rgb);
  return 0;
}

extern "C" int oracle_swap_in(oracle_engine *e, oracle_scene *s, oracle_render_state *r);
extern "C" int oracle_swap_out(oracle_engine *e, oracle_scene *s, oracle_render_state *r);

// denseMapper->ProcessFrame (InfiniTamDriver.h:187-192; DenseSlam.cpp:213,236,403)
extern "C" int oracle_process_frame(oracle_engine *e, oracle_scene *s, const oracle_view *v, oracle_render_state *r,
                         const float *M_d, const float *intr_d, const float *M_rgb, const float *intr_rgb,
                         int only_update_visible_list, int is_defusion) {
  oracle_allocate_scene_from_depth(e, s, v, r, M_d, intr_d, only_update_visible_list);
  integrate_impl<false>(e, s, v, r, M_d, intr_d, M_rgb, intr_rgb);
  push_visible_list(s, r, is_defusion ? 1 : 0);
  if (s->p.use_swapping) {
    oracle_swap_in(e, s, r);
    oracle_swap_out(e, s, r);
  }
  return 0;
}

// denseMapper->DeProcessFrame (InfiniTamDriver.h:194-199; DenseSlam.cpp:393,425).  Formula is a design
// decision: visible-list-only pass at the old pose, then the inverse running average.
extern "C" int oracle_deprocess_frame(oracle_engine *e, oracle_scene *s, const oracle_view *v, oracle_render_state *r,
                           const float *M_d, const float *intr_d, const float *M_rgb, const float *intr_rgb) {
  oracle_allocate_scene_from_depth(e, s, v, r, M_d, intr_d, 1);
  integrate_impl<true>(e, s, v, r, M_d, intr_d, M_rgb, intr_rgb);
  return 0;
}

// The blocks a keyframe was fused into, kept with it (same contract as dslam_frame_store_enable_lists /
// dslam_frame_store_put_visible_list / dslam_deprocess_frame_stored): de-integration visits exactly the listed entries
// that still hold the block they held at fusion time, and leaves the render state alone.
extern "C" int oracle_frame_store_enable_lists(oracle_engine *, oracle_frame_store *fs, const oracle_scene *) {
  if (!fs->lists_enabled) {
    fs->list_ids.assign(fs->capacity, {});
    fs->list_pos.assign(fs->capacity, {});
    fs->has_list.assign(fs->capacity, 0);
    fs->lists_enabled = true;
  }
  return 0;
}
extern "C" int oracle_frame_store_put_visible_list(oracle_engine *, oracle_frame_store *fs, int slot, const oracle_scene *s,
                                                   const oracle_render_state *r) {
  if (!fs->lists_enabled || slot < 0 || slot >= fs->capacity) return DSLAM_ERR_INVALID;
  fs->list_ids[slot].assign(r->visible_ids.begin(), r->visible_ids.begin() + r->no_visible);
  fs->list_pos[slot].resize(r->no_visible);
  for (int i = 0; i < r->no_visible; i++) {
    const dslam_hash_entry &he = s->hash[r->visible_ids[i]];
    fs->list_pos[slot][i] = S4{he.pos[0], he.pos[1], he.pos[2], 0};
  }
  fs->has_list[slot] = 1;
  return 0;
}
extern "C" int oracle_deprocess_frame_stored(oracle_engine *e, oracle_scene *s, const oracle_view *v, const oracle_frame_store *fs,
                                             int slot, const float *M_d, const float *intr_d, const float *M_rgb,
                                             const float *intr_rgb) {
  if (!fs->lists_enabled || slot < 0 || slot >= fs->capacity || !fs->has_list[slot]) return DSLAM_ERR_INVALID;
  oracle_render_state tmp;
  tmp.no_visible = 0;
  for (size_t i = 0; i < fs->list_ids[slot].size(); i++) {
    const int t = fs->list_ids[slot][i];
    const dslam_hash_entry &he = s->hash[t];
    const S4 &ep = fs->list_pos[slot][i];
    if (he.pos[0] != ep.x || he.pos[1] != ep.y || he.pos[2] != ep.z) continue;  // the entry holds another block now
    tmp.visible_ids.push_back(t);
    tmp.no_visible++;
  }
  integrate_impl<true>(e, s, v, &tmp, M_d, intr_d, M_rgb, intr_rgb);
  return 0;
}

// The re-integration batch of DenseSlam::OnlineCorrection (reference DenseSlam.cpp:389-403) as the reference runs it: keyframe
// by keyframe, DeProcessFrame at the old pose (here from the stored list), ProcessFrame(isDefusion) at the new one.  This
// sequence DEFINES dslam_reintegrate_batch; the HIP engine runs it block-major and must end in the same bytes.
extern "C" int oracle_reintegrate_batch(oracle_engine *e, oracle_scene *s, oracle_view *v, oracle_render_state *r, oracle_frame_store *fs,
                                        int n, const int32_t *slots, const float *old_M, const float *new_M, const float *intr,
                                        float affine_a, float affine_b) {
  for (int k = 0; k < n; k++) {
    int rc = oracle_view_update_from_store(e, v, fs, slots[k], affine_a, affine_b, 0.0, 0);
    if (rc) return rc;
    if ((rc = oracle_deprocess_frame_stored(e, s, v, fs, slots[k], old_M + 16 * (size_t)k, intr, nullptr, nullptr))) return rc;
    if ((rc = oracle_process_frame(e, s, v, r, new_M + 16 * (size_t)k, intr, nullptr, nullptr, 0, 1))) return rc;
    if ((rc = oracle_frame_store_put_visible_list(e, fs, slots[k], s, r))) return rc;
  }
  return 0;
}

// -------------------------------------------------------------------------------------------------
// Decay / SlideWindow (InfiniTamDriver.h:274-331).  The fork's bodies are not knowable;
// the semantics below are this build's documented design (DESIGN.md "Decay and sliding window").
// -------------------------------------------------------------------------------------------------
namespace {

// zero weak voxels of entry t's block; returns true if the block is now empty
static bool decay_block(oracle_scene *s, int t, int max_weight) {
  dslam_voxel *vb = &s->vba[(size_t)s->hash[t].ptr * 512];
  int non_empty = 0;
  for (int i = 0; i < 512; i++) {
    if (vb[i].w_depth > 0 && vb[i].w_depth <= max_weight) vb[i] = kEmptyVoxel;
    if (vb[i].w_depth > 0) non_empty++;
  }
  return non_empty == 0;
}

// decay the candidate entries (ascending index); empty blocks are released unless the scene swaps
// (ITMGlobalCache is indexed by hash entry, so entries of a swapping scene are never unlinked)
static void decay_candidates(oracle_scene *s, oracle_render_state *r, const std::vector<int> &cand, int max_weight) {
  std::vector<int> rem;
  for (int t : cand)
    if (decay_block(s, t, max_weight)) rem.push_back(t);
  if (s->p.use_swapping) return;
  s->decayed_blocks += (int64_t)rem.size();
  remove_entries(s, r, rem);
}

static int decay_impl(oracle_scene *s, oracle_render_state *r, int max_weight, int min_age, int force_all, int q) {
  if (!force_all) {
    // aged-list mode (DynSLAM PartialDecay): every queued list is decayed once, when it is min_age lists old
    const int bits = history_bits(s);
    const int newest = s->ring_next[q] - 1;
    int k = std::max(s->decay_cursor[q], s->ring_head[q]);
    for (; k <= newest - min_age; k++) {
      const int b = k % bits;
      std::vector<int> cand;
      for (int t = 0; t < s->n_entries; t++) {
        const int ptr = s->hash[t].ptr;
        if (ptr < 0) continue;
        if (slot_mask(s, ptr, q)[b >> 6] & (1ull << (b & 63))) cand.push_back(t);
      }
      decay_candidates(s, r, cand, max_weight);
    }
    if (k > s->decay_cursor[q]) s->decay_cursor[q] = k;
  } else {
    // full-sweep mode: every resident block not seen for min_age lists, once per observation epoch
    const int threshold = (s->frame_counter - 1) - min_age;
    std::vector<int> cand;
    for (int t = 0; t < s->n_entries; t++) {
      const int ptr = s->hash[t].ptr;
      if (ptr < 0) continue;
      const int ls = s->last_seen[ptr];
      if (ls < 0 || ls > threshold) continue;
      s->last_seen[ptr] = -2 - ls;
      cand.push_back(t);
    }
    decay_candidates(s, r, cand, max_weight);
  }
  return 0;
}

// ITMGlobalCache::SetStoredData for one block
static inline void store_block(oracle_scene *s, int t, const dslam_voxel *src) {
  memcpy(s->stored + (size_t)t * 512, src, 512 * sizeof(dslam_voxel));
  s->has_stored[t] = 1;
}

// CombineVoxelInformation (upstream ITMSwappingEngine.h): merge src (host) into dst (device)
static inline void combine_voxel(const dslam_voxel &src, dslam_voxel &dst, int maxW) {
  {
    int newW = dst.w_depth, oldW = src.w_depth;
    float newF = sdf_to_float(dst.sdf), oldF = sdf_to_float(src.sdf);
    if (oldW != 0) {
      newF = (float)oldW * oldF + (float)newW * newF;
      newW = oldW + newW;
      newF /= (float)newW;
      newW = std::min(newW, maxW);
      dst.w_depth = (uint8_t)newW;
      dst.sdf = float_to_sdf(newF);
    }
  }
  {
    int newW = dst.w_color, oldW = src.w_color;
    if (oldW != 0) {
      int sumW = oldW + newW;
      for (int k = 0; k < 3; k++) {
        float nc = (float)dst.clr[k] / 255.0f, oc = (float)src.clr[k] / 255.0f;
        nc = oc * (float)oldW + nc * (float)newW;
        nc /= (float)sumW;
        dst.clr[k] = (uint8_t)(nc * 255.0f);
      }
      dst.w_color = (uint8_t)std::min(sumW, maxW);
    }
  }
}

// pop the oldest list of ring q: blocks that no queued list references any more leave the device --
// released, or (scene with swapping, BASELINE config 2) moved to the host store with their entry kept.
static void slide_pop(oracle_scene *s, oracle_render_state *r, int q) {
  const int bits = history_bits(s);
  const int b = (s->ring_head[q]++) % bits;
  if (s->decay_cursor[q] < s->ring_head[q]) s->decay_cursor[q] = s->ring_head[q];
  std::vector<int> rem;
  for (int t = 0; t < s->n_entries; t++) {
    const int ptr = s->hash[t].ptr;
    if (ptr < 0) continue;
    uint64_t &w = slot_mask(s, ptr, q)[b >> 6];
    if (!(w & (1ull << (b & 63)))) continue;
    w &= ~(1ull << (b & 63));
    if (!slot_referenced(s, ptr)) rem.push_back(t);
  }
  s->slid_blocks += (int64_t)rem.size();
  if (!s->p.use_swapping) { remove_entries(s, r, rem); return; }
  for (int t : rem) {
    const int ptr = s->hash[t].ptr;
    dslam_voxel *vb = &s->vba[(size_t)ptr * 512];
    if (s->swap_state[t] != 2 && s->has_stored[t]) {  // host copy not merged yet: merge before storing
      const dslam_voxel *src = s->stored + (size_t)t * 512;
      for (int i = 0; i < 512; i++) combine_voxel(src[i], vb[i], s->p.max_w);
    }
    store_block(s, t, vb);
    for (int i = 0; i < 512; i++) vb[i] = kEmptyVoxel;
    s->alloc_list[++s->last_free] = ptr;
    slot_forget(s, ptr);
    s->hash[t].ptr = -1;
    s->swap_state[t] = 0;
    if (r) r->visible_type[t] = 0;
  }
  if (r && !rem.empty()) {
    int n = 0;
    for (int t = 0; t < r->n_entries; t++) if (r->visible_type[t] > 0) push_visible(r, n, t);
    r->no_visible = std::min(n, r->n_local);
  }
}

}  // namespace

extern "C" int oracle_decay(oracle_engine *, oracle_scene *s, oracle_render_state *r, int max_weight, int min_age, int force_all) {
  return decay_impl(s, r, max_weight, min_age, force_all, 0);
}
extern "C" int oracle_decay_defusion_part(oracle_engine *, oracle_scene *s, oracle_render_state *r, int max_weight, int min_age,
                               int force_all) {
  return decay_impl(s, r, max_weight, min_age, force_all, 1);
}
extern "C" int oracle_slide_window(oracle_engine *, oracle_scene *s, oracle_render_state *r, int max_age) {
  if (max_age < 0) max_age = 0;
  while (s->ring_next[0] - s->ring_head[0] > max_age) slide_pop(s, r, 0);
  return 0;
}
extern "C" int oracle_slide_window_defusion_part(oracle_engine *, oracle_scene *s, oracle_render_state *r, int max_age, int max_size) {
  (void)max_age;
  if (max_size < 0) max_size = 0;
  while (s->ring_next[1] - s->ring_head[1] > max_size) slide_pop(s, r, 1);
  return 0;
}

// -------------------------------------------------------------------------------------------------
// Swapping (ITMSwappingEngine, InfiniTamDriver.h:240-242, DenseSlam.h:248-251)
// -------------------------------------------------------------------------------------------------
extern "C" int oracle_swap_in(oracle_engine *, oracle_scene *s, oracle_render_state *) {
  if (!s->p.use_swapping) return DSLAM_ERR_INVALID;
  std::vector<int> needed;
  for (int t = 0; t < s->n_entries; t++) {
    if ((int)needed.size() >= DSLAM_TRANSFER_BLOCK_NUM) break;
    if (s->swap_state[t] == 1) needed.push_back(t);
  }
  for (int t : needed) {
    if (s->has_stored[t] && s->hash[t].ptr >= 0) {
      const dslam_voxel *src = s->stored + (size_t)t * 512;
      dslam_voxel *dst = &s->vba[(size_t)s->hash[t].ptr * 512];
      for (int i = 0; i < 512; i++) combine_voxel(src[i], dst[i], s->p.max_w);
    }
    s->swap_state[t] = 2;
  }
  s->last_swapped_in = (int)needed.size();
  return 0;
}

namespace {
static int swap_out_impl(oracle_scene *s, const uint8_t *visible_type) {
  int n = 0;
  for (int t = 0; t < s->n_entries; t++) {
    if (n >= DSLAM_TRANSFER_BLOCK_NUM) break;
    int ptr = s->hash[t].ptr;
    if (s->swap_state[t] == 2 && ptr >= 0 && (visible_type == nullptr || visible_type[t] == 0)) {
      dslam_voxel *vb = &s->vba[(size_t)ptr * 512];
      store_block(s, t, vb);
      s->swap_state[t] = 0;
      if (s->last_free < s->p.num_local_blocks - 1) {
        s->alloc_list[++s->last_free] = ptr;
        slot_forget(s, ptr);
        s->hash[t].ptr = -1;
        for (int i = 0; i < 512; i++) vb[i] = kEmptyVoxel;
      }
      n++;
    }
  }
  return n;
}
}  // namespace

extern "C" int oracle_swap_out(oracle_engine *, oracle_scene *s, oracle_render_state *r) {
  if (!s->p.use_swapping) return DSLAM_ERR_INVALID;
  s->last_swapped_out = swap_out_impl(s, r->visible_type.data());
  return 0;
}

// Hansry's SaveToGlobalMemory(scene) (DenseSlam.h:248-251): merge everything pending, then flush every
// resident block to the host store (design decision, SURVEY A.8 last line).
extern "C" int oracle_save_to_global_memory(oracle_engine *e, oracle_scene *s) {
  if (!s->p.use_swapping) return DSLAM_ERR_INVALID;
  while (true) { oracle_swap_in(e, s, nullptr); if (s->last_swapped_in == 0) break; }
  // entries that were never visible since allocation keep state 0 with resident data: promote them
  for (int t = 0; t < s->n_entries; t++) if (s->hash[t].ptr >= 0 && s->swap_state[t] == 0) {
    if (s->has_stored[t]) {
      const dslam_voxel *src = s->stored + (size_t)t * 512;
      dslam_voxel *dst = &s->vba[(size_t)s->hash[t].ptr * 512];
      for (int i = 0; i < 512; i++) combine_voxel(src[i], dst[i], s->p.max_w);
    }
    s->swap_state[t] = 2;
  }
  int total = 0;
  while (true) { int n = swap_out_impl(s, nullptr); total += n; if (n == 0) break; }
  s->last_swapped_out = total;
  return 0;
}

// -------------------------------------------------------------------------------------------------
// Visualisation (ITMMainEngine::GetImage, InfiniTamDriver.cpp:229-277)
// -------------------------------------------------------------------------------------------------
extern "C" int oracle_find_visible_blocks(oracle_engine *, const oracle_scene *s, oracle_render_state *r, const float *M,
                               const float *intr) {
  int n = 0;
  for (int t = 0; t < s->n_entries; t++) {
    const dslam_hash_entry &e = s->hash[t];
    bool vis = false, vis_enl = false;
    if (e.ptr >= 0) check_block_vis(vis, vis_enl, e.pos, M, intr, s->p.voxel_size, r->w, r->h, false);
    if (vis) push_visible(r, n, t);
  }
  r->no_visible = std::min(n, r->n_local);
  return 0;
}

extern "C" int oracle_count_visible_blocks(oracle_engine *, const oracle_scene *s, const oracle_render_state *r, int min_id,
                                int max_id, int *out) {
  int c = 0;
  for (int i = 0; i < r->no_visible; i++) {
    int ptr = s->hash[r->visible_ids[i]].ptr;
    if (ptr >= min_id && ptr <= max_id) c++;
  }
  *out = c;
  return 0;
}

namespace {
static const float FAR_AWAY = 999999.9f, VERY_CLOSE = 0.05f;

// float -> int with saturation (the GPU's v_cvt_i32_f32 saturates; x86 cvttss2si does not): only matters for
// block corners a few micrometres in front of the camera plane, whose projection overflows int
static inline int f2i_sat(float x) {
  if (x >= 2147483648.0f) return INT_MAX;
  if (x < -2147483648.0f) return INT_MIN;
  return (int)x;
}

static bool project_single_block(const int16_t *bp, const float *M, const float *intr, int W, int H, float vs,
                                 V2i &ul, V2i &lr, V2f &zr) {
  ul.x = W / 8; ul.y = H / 8;
  lr.x = -1; lr.y = -1;
  zr.x = FAR_AWAY; zr.y = VERY_CLOSE;
  for (int corner = 0; corner < 8; corner++) {
    int16_t tx = bp[0], ty = bp[1], tz = bp[2];
    tx += (corner & 1) ? 1 : 0; ty += (corner & 2) ? 1 : 0; tz += (corner & 4) ? 1 : 0;
    V4f p = {(float)tx * (float)DSLAM_BLOCK_SIZE * vs, (float)ty * (float)DSLAM_BLOCK_SIZE * vs,
             (float)tz * (float)DSLAM_BLOCK_SIZE * vs, 1.0f};
    p = mul(M, p);
    if (p.z < 1e-6f) continue;
    float px = (intr[0] * p.x / p.z + intr[2]) / 8.0f;
    float py = (intr[1] * p.y / p.z + intr[3]) / 8.0f;
    if ((float)ul.x > floorf(px)) ul.x = f2i_sat(floorf(px));
    if ((float)lr.x < ceilf(px)) lr.x = f2i_sat(ceilf(px));
    if ((float)ul.y > floorf(py)) ul.y = f2i_sat(floorf(py));
    if ((float)lr.y < ceilf(py)) lr.y = f2i_sat(ceilf(py));
    if (zr.x > p.z) zr.x = p.z;
    if (zr.y < p.z) zr.y = p.z;
  }
  if (ul.x < 0) ul.x = 0;
  if (ul.y < 0) ul.y = 0;
  if (lr.x >= W) lr.x = W - 1;
  if (lr.y >= H) lr.y = H - 1;
  if (ul.x > lr.x) return false;
  if (ul.y > lr.y) return false;
  if (zr.x < VERY_CLOSE) zr.x = VERY_CLOSE;
  if (zr.y < VERY_CLOSE) return false;
  return true;
}
}  // namespace

extern "C" int oracle_debug_set_render_tile_budget(oracle_engine *e, int budget) { e->render_tile_budget = budget; return 0; }

extern "C" int oracle_create_expected_depths(oracle_engine *eng, const oracle_scene *s, oracle_render_state *r, const float *M,
                                  const float *intr) {
  const int W = r->w, H = r->h;
  for (auto &px : r->range) { px.x = FAR_AWAY; px.y = VERY_CLOSE; }
  int num_rb = 0;
  for (int i = 0; i < r->no_visible; i++) {
    const dslam_hash_entry &e = s->hash[r->visible_ids[i]];
    V2i ul, lr; V2f zr;
    bool valid = false;
    if (e.ptr >= 0) valid = project_single_block(e.pos, M, intr, W, H, s->p.voxel_size, ul, lr, zr);
    if (!valid) continue;
    int rx = (int)ceilf((float)(lr.x - ul.x + 1) / 16.0f), ry = (int)ceilf((float)(lr.y - ul.y + 1) / 16.0f);
    int req = rx * ry;
    if (num_rb + req >= eng->render_tile_budget) continue;
    num_rb += req;
    // the render tiles partition the bbox exactly, so filling the bbox equals filling its tiles
    for (int y = ul.y; y <= lr.y; y++)
      for (int x = ul.x; x <= lr.x; x++) {
        V2f &px = r->range[x + (size_t)y * W];
        if (px.x > zr.x) px.x = zr.x;
        if (px.y < zr.y) px.y = zr.y;
      }
  }
  return 0;
}

namespace {
// debug counters (oracle_raycast_stats): ray-march steps, interpolated reads, rays
static long long g_dbg_steps = 0, g_dbg_interp = 0, g_dbg_rays = 0, g_dbg_maxsteps = 0;
static uint8_t *g_dbg_trace = nullptr;  // optional [W * H * g_dbg_trace_len]: kind of every march step of every ray (analysis)
static int g_dbg_trace_len = 0;
static int *g_dbg_pixel_steps = nullptr;  // optional [3 * W * H]: per ray (steps, steps that found no block, runs of such steps)

// castRay
static inline bool cast_ray(const oracle_scene *s, V4f &out, int x, int y, const float *invM, const float *intr,
                            float one_over_vs, float mu, const V2f &minmax, int dbg_loc = 0) {
  V4f pc; V3f ps, pe, dir, res;
  bool hash_found;
  float sdf = 1.0f;
  float total, step, total_max, step_scale;
  step_scale = mu * one_over_vs;
  const float inv_fx = 1.0f / intr[0], inv_fy = 1.0f / intr[1];

  pc.z = minmax.x;
  pc.x = pc.z * (((float)x - intr[2]) * inv_fx);
  pc.y = pc.z * (((float)y - intr[3]) * inv_fy);
  pc.w = 1.0f;
  total = sqrtf(pc.x * pc.x + pc.y * pc.y + pc.z * pc.z) * one_over_vs;
  V4f q = mul(invM, pc);
  ps.x = q.x * one_over_vs; ps.y = q.y * one_over_vs; ps.z = q.z * one_over_vs;

  pc.z = minmax.y;
  pc.x = pc.z * (((float)x - intr[2]) * inv_fx);
  pc.y = pc.z * (((float)y - intr[3]) * inv_fy);
  pc.w = 1.0f;
  total_max = sqrtf(pc.x * pc.x + pc.y * pc.y + pc.z * pc.z) * one_over_vs;
  q = mul(invM, pc);
  pe.x = q.x * one_over_vs; pe.y = q.y * one_over_vs; pe.z = q.z * one_over_vs;

  dir.x = pe.x - ps.x; dir.y = pe.y - ps.y; dir.z = pe.z - ps.z;
  float dn = 1.0f / sqrtf(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
  dir.x *= dn; dir.y *= dn; dir.z *= dn;
  res = ps;
  IndexCache cache;
  long long nsteps = 0, ninterp = 0, nmiss = 0, nruns = 0;
  bool prev_miss = false;
  V3i trace_key = {INT_MAX, INT_MAX, INT_MAX};  // analysis only: block of the last straddling cell's lower corner
  V3i trace_cache = {INT_MAX, INT_MAX, INT_MAX};
  while (total < total_max) {
    nsteps++;
    V3i probe_block = {0, 0, 0};
    bool probe_cached = false;
    if (g_dbg_trace) {  // analysis only; (the engine's per-ray block cache is refreshed by the nearest-voxel probe only)
      point_to_block(V3i{iround(res.x), iround(res.y), iround(res.z)}, probe_block);
      probe_cached = probe_block.x == trace_cache.x && probe_block.y == trace_cache.y && probe_block.z == trace_cache.z;
    }
    sdf = read_sdf_uninterp(s, res, hash_found, cache);
    if (g_dbg_trace && hash_found) trace_cache = probe_block;
    uint8_t kind = 0;
    if (!hash_found) {
      nmiss++;
      if (!prev_miss) nruns++;
      prev_miss = true;
      step = (float)DSLAM_BLOCK_SIZE;
      kind = 1;
      if (g_dbg_trace && nsteps <= g_dbg_trace_len) g_dbg_trace[(size_t)dbg_loc * g_dbg_trace_len + nsteps - 1] = kind;
    } else {
      prev_miss = false;
      kind = probe_cached ? 2 : 3;
      if ((sdf <= 0.1f) && (sdf >= -0.5f)) {
        if (g_dbg_trace) {  // does the trilinear cell straddle blocks, and is it the neighbourhood of the last such cell?
          const int x0 = (int)floorf(res.x), y0 = (int)floorf(res.y), z0 = (int)floorf(res.z);
          V3i lo, hi;
          point_to_block(V3i{x0, y0, z0}, lo);
          point_to_block(V3i{x0 + 1, y0 + 1, z0 + 1}, hi);
          if (lo.x != hi.x || lo.y != hi.y || lo.z != hi.z) {
            const bool same = lo.x == trace_key.x && lo.y == trace_key.y && lo.z == trace_key.z;
            kind = (probe_cached ? 4 : 6) + (same ? 1 : 0);  // 4/5: straddling (new / same neighbourhood), 6/7 with a probe
            trace_key = lo;
          }
        }
        sdf = read_sdf_interp(s, res, hash_found, cache);
        ninterp++;
      }
      if (g_dbg_trace && nsteps <= g_dbg_trace_len) g_dbg_trace[(size_t)dbg_loc * g_dbg_trace_len + nsteps - 1] = kind;
      if (sdf <= 0.0f) break;
      step = std::max(sdf * step_scale, 1.0f);
    }
    res.x += step * dir.x; res.y += step * dir.y; res.z += step * dir.z;
    total += step;
  }
  bool pt_found;
#pragma omp atomic
  g_dbg_steps += nsteps;
#pragma omp atomic
  g_dbg_interp += ninterp;
#pragma omp atomic
  g_dbg_rays += 1;
  if (nsteps > g_dbg_maxsteps) g_dbg_maxsteps = nsteps;
  if (g_dbg_pixel_steps) { g_dbg_pixel_steps[3 * dbg_loc] = nsteps; g_dbg_pixel_steps[3 * dbg_loc + 1] = nmiss; g_dbg_pixel_steps[3 * dbg_loc + 2] = nruns; }
  if (sdf <= 0.0f) {
    step = sdf * step_scale;
    res.x += step * dir.x; res.y += step * dir.y; res.z += step * dir.z;
    sdf = read_sdf_interp(s, res, hash_found, cache);
    step = sdf * step_scale;
    res.x += step * dir.x; res.y += step * dir.y; res.z += step * dir.z;
    pt_found = true;
  } else pt_found = false;
  out.x = res.x; out.y = res.y; out.z = res.z; out.w = pt_found ? 1.0f : 0.0f;
  return pt_found;
}

static void generic_raycast(oracle_engine *e, const oracle_scene *s, oracle_render_state *r, const float *invM,
                            const float *intr) {
  const int W = r->w, H = r->h;
  const float one_over_vs = 1.0f / s->p.voxel_size;
#pragma omp parallel for schedule(dynamic, 256) if (e->threads > 1)
  for (int loc = 0; loc < W * H; loc++) {
    int y = loc / W, x = loc - y * W;
    int loc2 = (int)floorf((float)x / 8.0f) + (int)floorf((float)y / 8.0f) * W;
    cast_ray(s, r->raycast[loc], x, y, invM, intr, one_over_vs, s->p.mu, r->range[loc2], loc);
  }
}

static inline void normal_and_angle(const oracle_scene *s, bool &found, const V3f &pt, const V3f &light, V3f &n,
                                    float &angle, IndexCache &c) {
  if (!found) return;
  n = normal_from_sdf(s, pt, c);
  float ns = 1.0f / sqrtf(n.x * n.x + n.y * n.y + n.z * n.z);
  n.x *= ns; n.y *= ns; n.z *= ns;
  angle = n.x * light.x + n.y * light.y + n.z * light.z;
  if (!(angle > 0.0f)) found = false;
}
}  // namespace

// RenderImage (raycast + shading).  DSLAM_IMAGE_DEPTH = Hansry's FREECAMERA_DEPTH; its definition is a
// design decision: camera-frame z of the hit point in metres, 0 where nothing was hit.
extern "C" int oracle_render_image(oracle_engine *e, const oracle_scene *s, oracle_render_state *r, const float *M,
                        const float *intr, int type, uint8_t *out_rgba, float *out_float) {
  const int W = r->w, H = r->h;
  float invM[16];
  inv4(M, invM);
  generic_raycast(e, s, r, invM, intr);
  V3f light = {-invM[8], -invM[9], -invM[10]};  // -invM.getColumn(2)
  const float vs = s->p.voxel_size;
#pragma omp parallel for schedule(dynamic, 256) if (e->threads > 1)
  for (int loc = 0; loc < W * H; loc++) {
    V4f pr = r->raycast[loc];
    V3f pt = {pr.x, pr.y, pr.z};
    bool found = pr.w > 0;
    IndexCache c;
    uint8_t *o = &r->image_rgba[(size_t)loc * 4];
    if (type == DSLAM_IMAGE_DEPTH) {
      float d = 0.0f;
      if (found) {
        V4f pw = {pt.x * vs, pt.y * vs, pt.z * vs, 1.0f};
        d = mul(M, pw).z;
      }
      r->image_float[loc] = d;
      continue;
    }
    V3f n = {0, 0, 0}; float angle = 0.0f;
    normal_and_angle(s, found, pt, light, n, angle, c);
    if (!found) { o[0] = o[1] = o[2] = o[3] = 0; continue; }
    if (type == DSLAM_IMAGE_COLOUR_FROM_VOLUME) {
      V4f clr = read_colour_interp(s, pt, c);
      o[0] = (uint8_t)(clr.x * 255.0f); o[1] = (uint8_t)(clr.y * 255.0f); o[2] = (uint8_t)(clr.z * 255.0f); o[3] = 255;
    } else if (type == DSLAM_IMAGE_COLOUR_FROM_NORMAL) {
      o[0] = (uint8_t)((0.3f + (-n.x + 1.0f) * 0.35f) * 255.0f);
      o[1] = (uint8_t)((0.3f + (-n.y + 1.0f) * 0.35f) * 255.0f);
      o[2] = (uint8_t)((0.3f + (-n.z + 1.0f) * 0.35f) * 255.0f);
      o[3] = 255;
    } else {
      float g = (0.8f * angle + 0.2f) * 255.0f;
      o[0] = o[1] = o[2] = o[3] = (uint8_t)g;
    }
  }
  if (out_rgba) memcpy(out_rgba, r->image_rgba.data(), r->image_rgba.size());
  if (out_float) memcpy(out_float, r->image_float.data(), r->image_float.size() * sizeof(float));
  return 0;
}

extern "C" int oracle_get_image(oracle_engine *e, const oracle_scene *s, oracle_render_state *r, const float *M, const float *intr,
                     int type, uint8_t *out_rgba, float *out_float) {
  oracle_find_visible_blocks(e, s, r, M, intr);
  oracle_create_expected_depths(e, s, r, M, intr);
  return oracle_render_image(e, s, r, M, intr, type, out_rgba, out_float);
}

// GetImage(FREECAMERA_DEPTH) + FloatDepthmapToShort / FloatDepthmapToInt16 (InfiniTamDriver.cpp:167-200)
extern "C" int oracle_get_image(oracle_engine *e, const oracle_scene *s, oracle_render_state *r, const float *M, const float *intr,
                     int type, uint8_t *out_rgba, float *out_float);
extern "C" int oracle_get_depth_image_int16(oracle_engine *e, const oracle_scene *s, oracle_render_state *r, const float *M,
                                 const float *intr, int scale, int16_t *out) {
  std::vector<float> d((size_t)r->w * r->h);
  int rc = oracle_get_image(e, s, r, M, intr, DSLAM_IMAGE_DEPTH, nullptr, d.data());
  if (rc) return rc;
  for (size_t i = 0; i < d.size(); i++) out[i] = wrap_i16(d[i] * (float)scale);
  return 0;
}

// trackingController->Prepare -> CreateICPMaps (InfiniTamDriver.h:208-220): processPixelICP<true,false>
extern "C" int oracle_create_icp_maps(oracle_engine *e, const oracle_scene *s, oracle_render_state *r, const float *M,
                           const float *intr, float *out_points, float *out_normals) {
  const int W = r->w, H = r->h;
  float invM[16];
  inv4(M, invM);
  r->icp_points.resize((size_t)W * H);
  r->icp_normals.resize((size_t)W * H);
  r->raycast_image.resize((size_t)W * H * 4);
  oracle_create_expected_depths(e, s, r, M, intr);
  generic_raycast(e, s, r, invM, intr);
  V3f light = {-invM[8], -invM[9], -invM[10]};
  const float vs = s->p.voxel_size;
  const V4f *pr = r->raycast.data();
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      int loc = x + y * W;
      V4f point = pr[loc];
      bool found = point.w > 0.0f;
      V3f n = {0, 0, 0};
      float angle = 0.0f;
      if (found) {
        if (y <= 2 || y >= H - 3 || x <= 2 || x >= W - 3) found = false;
      }
      if (found) {
        V4f xp = pr[(x + 2) + y * W], yp = pr[x + (y + 2) * W], xm = pr[(x - 2) + y * W], ym = pr[x + (y - 2) * W];
        V4f dx = {0, 0, 0, 0}, dy = {0, 0, 0, 0};
        bool plus1 = false;
        if (xp.w <= 0 || yp.w <= 0 || xm.w <= 0 || ym.w <= 0) plus1 = true;
        if (!plus1) {
          dx = V4f{xp.x - xm.x, xp.y - xm.y, xp.z - xm.z, xp.w - xm.w};
          dy = V4f{yp.x - ym.x, yp.y - ym.y, yp.z - ym.z, yp.w - ym.w};
          float ld = std::max(dx.x * dx.x + dx.y * dx.y + dx.z * dx.z, dy.x * dy.x + dy.y * dy.y + dy.z * dy.z);
          if (ld * vs * vs > (0.15f * 0.15f)) plus1 = true;
        }
        if (plus1) {
          xp = pr[(x + 1) + y * W]; yp = pr[x + (y + 1) * W]; xm = pr[(x - 1) + y * W]; ym = pr[x + (y - 1) * W];
          dx = V4f{xp.x - xm.x, xp.y - xm.y, xp.z - xm.z, xp.w - xm.w};
          dy = V4f{yp.x - ym.x, yp.y - ym.y, yp.z - ym.z, yp.w - ym.w};
          if (xp.w <= 0 || yp.w <= 0 || xm.w <= 0 || ym.w <= 0) found = false;
        }
        if (found) {
          n.x = -(dx.y * dy.z - dx.z * dy.y);
          n.y = -(dx.z * dy.x - dx.x * dy.z);
          n.z = -(dx.x * dy.y - dx.y * dy.x);
          float ns = 1.0f / sqrtf(n.x * n.x + n.y * n.y + n.z * n.z);
          n.x *= ns; n.y *= ns; n.z *= ns;
          angle = n.x * light.x + n.y * light.y + n.z * light.z;
          if (!(angle > 0.0f)) found = false;
        }
      }
      // drawPixelGrey into renderState->raycastImage (what InfiniTAM_IMAGE_SCENERAYCAST shows, InfiniTamDriver.cpp:28-29)
      const uint8_t g = found ? (uint8_t)((0.8f * angle + 0.2f) * 255.0f) : (uint8_t)0;
      uint8_t *o = &r->raycast_image[(size_t)loc * 4];
      o[0] = o[1] = o[2] = o[3] = g;
      V4f pv, nv;
      if (found) {
        pv = V4f{point.x * vs, point.y * vs, point.z * vs, 1.0f};
        nv = V4f{n.x, n.y, n.z, 0.0f};
      } else {
        pv = V4f{0.0f, 0.0f, 0.0f, -1.0f};
        nv = V4f{0.0f, 0.0f, 0.0f, -1.0f};
      }
      r->icp_points[loc] = pv;
      r->icp_normals[loc] = nv;
      if (out_points) memcpy(out_points + (size_t)loc * 4, &pv, 16);
      if (out_normals) memcpy(out_normals + (size_t)loc * 4, &nv, 16);
    }
  return 0;
}

extern "C" int oracle_download_raycast_image(oracle_engine *, const oracle_render_state *r, uint8_t *out) {
  if (r->raycast_image.empty()) return DSLAM_ERR_INVALID;
  memcpy(out, r->raycast_image.data(), r->raycast_image.size());
  return 0;
}

// -------------------------------------------------------------------------------------------------
// ITMDepthTracker::TrackCamera (upstream InfiniTAM v2, [UPSTREAM-RECALL]; reached from
// trackingController->Track, InfiniTamDriver.h:151-163)
// -------------------------------------------------------------------------------------------------
namespace {
// FilterSubsampleWithHoles: average of the valid (> 0) pixels of each 2x2 group
void subsample_with_holes(const std::vector<float> &in, int w, int h, std::vector<float> &out) {
  const int nw = w / 2, nh = h / 2;
  out.assign((size_t)nw * nh, 0.0f);
  for (int y = 0; y < nh; y++)
    for (int x = 0; x < nw; x++) {
      float acc = 0.0f, good = 0.0f;
      for (int dy = 0; dy < 2; dy++)
        for (int dx = 0; dx < 2; dx++) {
          const float p = in[(2 * x + dx) + (size_t)(2 * y + dy) * w];
          if (p > 0.0f) { acc += p; good++; }
        }
      if (good > 0) acc /= good;
      out[x + (size_t)y * nw] = acc;
    }
}

inline V4f bilinear_with_holes(const V4f *src, float px, float py, int w) {
  const int ix = (short)floorf(px), iy = (short)floorf(py);
  const float dx = px - (float)ix, dy = py - (float)iy;
  const V4f a = src[ix + (size_t)iy * w], b = src[(ix + 1) + (size_t)iy * w];
  const V4f c = src[ix + (size_t)(iy + 1) * w], d = src[(ix + 1) + (size_t)(iy + 1) * w];
  if (a.w < 0 || b.w < 0 || c.w < 0 || d.w < 0) return V4f{0, 0, 0, -1.0f};
  V4f r;
  r.x = a.x * (1.0f - dx) * (1.0f - dy) + b.x * dx * (1.0f - dy) + c.x * (1.0f - dx) * dy + d.x * dx * dy;
  r.y = a.y * (1.0f - dx) * (1.0f - dy) + b.y * dx * (1.0f - dy) + c.y * (1.0f - dx) * dy + d.y * dx * dy;
  r.z = a.z * (1.0f - dx) * (1.0f - dy) + b.z * dx * (1.0f - dy) + c.z * (1.0f - dx) * dy + d.z * dx * dy;
  r.w = a.w * (1.0f - dx) * (1.0f - dy) + b.w * dx * (1.0f - dy) + c.w * (1.0f - dx) * dy + d.w * dx * dy;
  return r;
}

// computePerPointGH_Depth_Ab: the point-to-plane residual b and its Jacobian row A (3 or 6 entries)
inline bool per_point_ab(float *A, float &b, int x, int y, float depth, const float *view_intr, int sw, int sh,
                         const float *scene_intr, const float *approxInvPose, const float *scenePose,
                         const V4f *points, const V4f *normals, float dist_thresh, int type) {
  if (depth <= 1e-8f) return false;
  V4f p;
  p.x = depth * (((float)x - view_intr[2]) / view_intr[0]);
  p.y = depth * (((float)y - view_intr[3]) / view_intr[1]);
  p.z = depth;
  p.w = 1.0f;
  p = mul(approxInvPose, p);
  p.w = 1.0f;
  const V4f q = mul(scenePose, p);
  if (q.z <= 0.0f) return false;
  const float u = scene_intr[0] * q.x / q.z + scene_intr[2], v = scene_intr[1] * q.y / q.z + scene_intr[3];
  if (!((u >= 0.0f) && (u <= (float)(sw - 2)) && (v >= 0.0f) && (v <= (float)(sh - 2)))) return false;
  const V4f cp = bilinear_with_holes(points, u, v, sw);
  if (cp.w < 0.0f) return false;
  const float ddx = cp.x - p.x, ddy = cp.y - p.y, ddz = cp.z - p.z;
  const float dist = ddx * ddx + ddy * ddy + ddz * ddz;
  if (dist > dist_thresh) return false;
  const V4f n = bilinear_with_holes(normals, u, v, sw);
  b = n.x * ddx + n.y * ddy + n.z * ddz;
  const float r0 = +p.z * n.y - p.y * n.z, r1 = -p.z * n.x + p.x * n.z, r2 = +p.y * n.x - p.x * n.y;
  if (type == DSLAM_TRACKER_ITERATION_ROTATION) { A[0] = r0; A[1] = r1; A[2] = r2; }
  else if (type == DSLAM_TRACKER_ITERATION_TRANSLATION) { A[0] = n.x; A[1] = n.y; A[2] = n.z; }
  else { A[0] = r0; A[1] = r1; A[2] = r2; A[3] = n.x; A[4] = n.y; A[5] = n.z; }
  return true;
}

// Cholesky solve of the (damped) normal equations, float like ORUtils::Cholesky
void cholesky_solve(const float *Ain, int n, const float *bvec, float *x) {
  float L[36];
  for (int i = 0; i < n * n; i++) L[i] = Ain[i];
  for (int c = 0; c < n; c++) {
    float inv_diag = 1.0f;
    for (int r = c; r < n; r++) {
      float val = L[c + r * n];
      for (int c2 = 0; c2 < c; c2++) val -= L[c + c2 * n] * L[c2 + r * n];
      if (r == c) { L[c + r * n] = val; inv_diag = (val == 0.0f) ? 0.0f : 1.0f / val; }
      else { L[r + c * n] = val; L[c + r * n] = val * inv_diag; }
    }
  }
  float yv[6];
  for (int i = 0; i < n; i++) {
    float val = bvec[i];
    for (int j = 0; j < i; j++) val -= L[j + i * n] * yv[j];
    yv[i] = val;
  }
  for (int i = 0; i < n; i++) yv[i] = (L[i + i * n] == 0.0f) ? 0.0f : yv[i] / L[i + i * n];
  for (int i = n - 1; i >= 0; i--) {
    float val = yv[i];
    for (int j = i + 1; j < n; j++) val -= L[i + j * n] * x[j];
    x[i] = val;
  }
}

// ITMPose::Coerce stand-in: Gram-Schmidt on the rotation columns, bottom row (0,0,0,1)
void coerce_pose(float *M) {
  double c0[3] = {M[0], M[1], M[2]}, c1[3] = {M[4], M[5], M[6]}, c2[3];
  double n = sqrt(c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2]);
  if (n > 0) for (double &v : c0) v /= n;
  const double d = c0[0] * c1[0] + c0[1] * c1[1] + c0[2] * c1[2];
  for (int i = 0; i < 3; i++) c1[i] -= d * c0[i];
  n = sqrt(c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2]);
  if (n > 0) for (double &v : c1) v /= n;
  c2[0] = c0[1] * c1[2] - c0[2] * c1[1]; c2[1] = c0[2] * c1[0] - c0[0] * c1[2]; c2[2] = c0[0] * c1[1] - c0[1] * c1[0];
  for (int i = 0; i < 3; i++) { M[i] = (float)c0[i]; M[4 + i] = (float)c1[i]; M[8 + i] = (float)c2[i]; }
  M[3] = M[7] = M[11] = 0.0f; M[15] = 1.0f;
}
}  // namespace

extern "C" int oracle_track_camera(oracle_engine *, const oracle_view *v, oracle_render_state *r, const float *scenePose,
                        float *pose_M, const float *intr, const dslam_tracker_params *tp, dslam_tracker_result *res) {
  const int levels = tp->no_hierarchy_levels;
  if (levels < 1 || levels > DSLAM_TRACKER_MAX_LEVELS || tp->no_icp_run_till_level < 0) return DSLAM_ERR_INVALID;
  if (r->icp_points.size() != (size_t)r->w * r->h || v->w_d != r->w || v->h_d != r->h) return DSLAM_ERR_INVALID;
  // view hierarchy (PrepareForEvaluation); the scene maps stay at level 0
  std::vector<std::vector<float>> depth(levels);
  std::vector<int> lw(levels), lh(levels);
  float lintr[DSLAM_TRACKER_MAX_LEVELS][4];
  depth[0] = v->depth; lw[0] = v->w_d; lh[0] = v->h_d;
  for (int k = 0; k < 4; k++) lintr[0][k] = intr[k];
  for (int i = 1; i < levels; i++) {
    subsample_with_holes(depth[i - 1], lw[i - 1], lh[i - 1], depth[i]);
    lw[i] = lw[i - 1] / 2; lh[i] = lh[i - 1] / 2;
    for (int k = 0; k < 4; k++) lintr[i][k] = lintr[i - 1][k] * 0.5f;
  }
  int iters_per_level[DSLAM_TRACKER_MAX_LEVELS];
  float dist_per_level[DSLAM_TRACKER_MAX_LEVELS];
  iters_per_level[0] = 2;
  for (int i = 1; i < levels; i++) iters_per_level[i] = iters_per_level[i - 1] + 2;
  const float dstep = tp->dist_thresh / levels;
  dist_per_level[levels - 1] = tp->dist_thresh;
  for (int i = levels - 2; i >= 0; i--) dist_per_level[i] = dist_per_level[i + 1] - dstep;

  float M[16], approxInvPose[16];
  memcpy(M, pose_M, 64);
  float hessian_good[36] = {0}, nabla_good[6] = {0};
  int total_iters = 0, last_valid = 0;
  float last_f = 0.0f;
  for (int level = levels - 1; level >= tp->no_icp_run_till_level; level--) {
    const int type = tp->regime[level];
    if (type == DSLAM_TRACKER_ITERATION_NONE) continue;
    const int npara = (type == DSLAM_TRACKER_ITERATION_BOTH) ? 6 : 3;
    inv4(M, approxInvPose);
    float good_M[16];
    memcpy(good_M, M, 64);
    float f_old = 1e20f, lambda = 1.0f;
    for (int it = 0; it < iters_per_level[level]; it++) {
      // ComputeGandH: sums over the level's pixels; accumulated in double here and on the device (upstream adds
      // floats in pixel order, which no parallel machine reproduces) and rounded to float once
      double sumH[21] = {0}, sumN[6] = {0}, sumF = 0;
      int valid = 0;
      for (int y = 0; y < lh[level]; y++)
        for (int x = 0; x < lw[level]; x++) {
          float A[6], b;
          if (!per_point_ab(A, b, x, y, depth[level][x + (size_t)y * lw[level]], lintr[level], r->w, r->h, lintr[0],
                            approxInvPose, scenePose, r->icp_points.data(), r->icp_normals.data(), dist_per_level[level], type))
            continue;
          valid++;
          sumF += (double)(b * b);
          for (int k = 0, c = 0; k < npara; k++) {
            sumN[k] += (double)(b * A[k]);
            for (int j = 0; j <= k; j++, c++) sumH[c] += (double)(A[k] * A[j]);
          }
        }
      float hessian_new[36] = {0}, nabla_new[6] = {0};
      for (int k = 0, c = 0; k < npara; k++)
        for (int j = 0; j <= k; j++, c++) hessian_new[k + j * 6] = hessian_new[j + k * 6] = (float)sumH[c];
      for (int k = 0; k < npara; k++) nabla_new[k] = (float)sumN[k];
      const float f_new = (valid > 100) ? sqrtf((float)sumF) / (float)valid : 1e5f;
      total_iters++; last_valid = valid; last_f = f_new;

      if (valid <= 0 || f_new > f_old) {
        memcpy(M, good_M, 64);
        inv4(M, approxInvPose);
        lambda *= 10.0f;
      } else {
        memcpy(good_M, M, 64);
        f_old = f_new;
        for (int i = 0; i < 36; i++) hessian_good[i] = hessian_new[i] / (float)valid;
        for (int i = 0; i < 6; i++) nabla_good[i] = nabla_new[i] / (float)valid;
        lambda /= 10.0f;
      }
      float A6[36];
      for (int i = 0; i < 36; i++) A6[i] = hessian_good[i];
      for (int i = 0; i < 6; i++) A6[i + i * 6] *= 1.0f + lambda;
      float step[6] = {0, 0, 0, 0, 0, 0};
      if (npara == 3) {
        float small[9];
        for (int rr = 0; rr < 3; rr++) for (int cc = 0; cc < 3; cc++) small[rr + cc * 3] = A6[rr + cc * 6];
        cholesky_solve(small, 3, nabla_good, step);
      } else {
        cholesky_solve(A6, 6, nabla_good, step);
      }
      // ApplyDelta
      float s6[6] = {0, 0, 0, 0, 0, 0};
      if (type == DSLAM_TRACKER_ITERATION_ROTATION) { s6[0] = step[0]; s6[1] = step[1]; s6[2] = step[2]; }
      else if (type == DSLAM_TRACKER_ITERATION_TRANSLATION) { s6[3] = step[0]; s6[4] = step[1]; s6[5] = step[2]; }
      else for (int i = 0; i < 6; i++) s6[i] = step[i];
      // Tinc, column-major m[col * 4 + row]
      const float Tinc[16] = {1.0f, -s6[2], s6[1], 0.0f, s6[2], 1.0f, -s6[0], 0.0f, -s6[1], s6[0], 1.0f, 0.0f, s6[3], s6[4], s6[5], 1.0f};
      float next[16];
      for (int c = 0; c < 4; c++)
        for (int rr = 0; rr < 4; rr++) {
          float acc = 0;
          for (int k = 0; k < 4; k++) acc += Tinc[k * 4 + rr] * approxInvPose[c * 4 + k];
          next[c * 4 + rr] = acc;
        }
      inv4(next, M);       // pose_d->SetInvM(approxInvPose)
      coerce_pose(M);      // pose_d->Coerce()
      inv4(M, approxInvPose);
      float len = 0.0f;
      for (int i = 0; i < 6; i++) len += step[i] * step[i];
      if (sqrtf(len) / 6 < tp->termination_threshold) break;  // HasConverged
    }
  }
  memcpy(pose_M, M, 64);
  if (res) { res->iterations = total_iters; res->valid_points_last = last_valid; res->f_last = last_f; res->pad = 0; }
  return 0;
}

// -------------------------------------------------------------------------------------------------
// meshing export: ITMMeshingEngine_CPU::MeshScene (SaveCurrSceneToMesh, DenseSlam.cpp:638-643)
// upstream ITMLib/Engine/DeviceAgnostic/ITMMeshingEngine.h (findPointNeighbors, sdfInterp, buildVertList) and
// DeviceSpecific/CPU/ITMMeshingEngine_CPU.tpp, restated; sequential on purpose (the triangle order is the contract).
// -------------------------------------------------------------------------------------------------
namespace {

// readVoxel without a cache, as the meshing code calls it
static inline dslam_voxel read_voxel_uncached(const oracle_scene *s, const V3i &p, bool &found) {
  IndexCache fresh;
  return read_voxel(s, p, found, fresh);
}

// findPointNeighbors: the 8 corner samples of the cube at `base`; false when one is missing or has sdf == 1
static bool find_point_neighbours(const oracle_scene *s, const V3i &base, V3f *p, float *sdf, dslam_voxel *vox) {
  for (int k = 0; k < 8; k++) {
    const V3i q = {base.x + kMcCornerOffsets[k][0], base.y + kMcCornerOffsets[k][1], base.z + kMcCornerOffsets[k][2]};
    bool found;
    vox[k] = read_voxel_uncached(s, q, found);
    sdf[k] = sdf_to_float(vox[k].sdf);
    if (!found || sdf[k] == 1.0f) return false;
    p[k] = V3f{(float)q.x, (float)q.y, (float)q.z};
  }
  return true;
}

// sdfInterp, also applied to the colour channels (same early-outs, same weight)
static inline float interp_value(float a, float b, float v1, float v2) {
  if (fabsf(0.0f - v1) < 0.00001f) return a;
  if (fabsf(0.0f - v2) < 0.00001f) return b;
  if (fabsf(v1 - v2) < 0.00001f) return a;
  return a + ((0.0f - v1) / (v2 - v1)) * (b - a);
}

}  // namespace

extern "C" int oracle_mesh_scene(oracle_engine *e, const oracle_scene *s, int max_triangles, int with_colour, int *out_n) {
  if (max_triangles <= 0) max_triangles = s->p.num_local_blocks * 32;  // ITMMesh::noMaxTriangles
  const float factor = s->p.voxel_size;
  e->mesh_pos.clear();
  e->mesh_col.clear();
  int n = 0;
  float tri[9], col[9];
  for (int entry = 0; entry < s->n_entries; entry++) {
    const dslam_hash_entry &he = s->hash[entry];
    if (he.ptr < 0) continue;
    const V3i global = {he.pos[0] * DSLAM_BLOCK_SIZE, he.pos[1] * DSLAM_BLOCK_SIZE, he.pos[2] * DSLAM_BLOCK_SIZE};
    for (int z = 0; z < DSLAM_BLOCK_SIZE; z++)
      for (int y = 0; y < DSLAM_BLOCK_SIZE; y++)
        for (int x = 0; x < DSLAM_BLOCK_SIZE; x++) {
          V3f pts[8];
          float sdf[8];
          dslam_voxel vox[8];
          if (!find_point_neighbours(s, V3i{global.x + x, global.y + y, global.z + z}, pts, sdf, vox)) continue;
          int cube = 0;
          for (int k = 0; k < 8; k++)
            if (sdf[k] < 0) cube |= 1 << k;
          int edges = 0;  // upstream's edgeTable[cube]: the edges whose end corners differ in sign
          for (int ed = 0; ed < 12; ed++)
            if (((cube >> kMcEdgeCorners[ed][0]) & 1) != ((cube >> kMcEdgeCorners[ed][1]) & 1)) edges |= 1 << ed;
          if (edges == 0) continue;
          for (int i = 0; kMcTriangles[cube][i] != -1; i += 3) {
            for (int k = 0; k < 3; k++) {
              const int ed = kMcTriangles[cube][i + k];
              const int a = kMcEdgeCorners[ed][0], b = kMcEdgeCorners[ed][1];
              tri[3 * k + 0] = interp_value(pts[a].x, pts[b].x, sdf[a], sdf[b]) * factor;
              tri[3 * k + 1] = interp_value(pts[a].y, pts[b].y, sdf[a], sdf[b]) * factor;
              tri[3 * k + 2] = interp_value(pts[a].z, pts[b].z, sdf[a], sdf[b]) * factor;
              for (int ch = 0; ch < 3; ch++)
                col[3 * k + ch] = interp_value((float)vox[a].clr[ch], (float)vox[b].clr[ch], sdf[a], sdf[b]) / 255.0f;
            }
            // triangles[noTriangles] = t; if (noTriangles < noMaxTriangles - 1) noTriangles++;
            if (n < max_triangles - 1) {
              e->mesh_pos.insert(e->mesh_pos.end(), tri, tri + 9);
              if (with_colour) e->mesh_col.insert(e->mesh_col.end(), col, col + 9);
              n++;
            }
          }
        }
  }
  e->mesh_has_colour = with_colour != 0;
  *out_n = n;
  return 0;
}

extern "C" int oracle_mesh_download(oracle_engine *e, float *out_positions, float *out_colours, int capacity) {
  const int n = (int)(e->mesh_pos.size() / 9);
  if (capacity < n || (out_colours && !e->mesh_has_colour)) return DSLAM_ERR_INVALID;
  if (n) memcpy(out_positions, e->mesh_pos.data(), (size_t)n * 9 * sizeof(float));
  if (n && out_colours) memcpy(out_colours, e->mesh_col.data(), (size_t)n * 9 * sizeof(float));
  return 0;
}

// -------------------------------------------------------------------------------------------------
// read-back
// -------------------------------------------------------------------------------------------------
extern "C" int oracle_get_stats(oracle_engine *, const oracle_scene *s, const oracle_render_state *r, dslam_stats *o) {
  memset(o, 0, sizeof(*o));
  o->num_allocated_blocks = s->p.num_local_blocks;
  o->last_free_block_id = s->last_free;
  o->last_free_excess_id = s->last_free_ex;
  o->no_visible_entries = r ? r->no_visible : 0;
  o->decayed_block_count = s->decayed_blocks;
  o->slid_block_count = s->slid_blocks;
  o->frame_counter = s->frame_counter;
  o->fusion_fifo_len = s->ring_next[0] - s->ring_head[0];
  o->defusion_fifo_len = s->ring_next[1] - s->ring_head[1];
  o->alloc_failures = s->alloc_failures;
  o->last_swapped_in = s->last_swapped_in;
  o->last_swapped_out = s->last_swapped_out;
  return 0;
}
extern "C" int oracle_scene_get_params(const oracle_scene *s, dslam_scene_params *o) { *o = s->p; return 0; }
extern "C" int oracle_download_hash_table(oracle_engine *, const oracle_scene *s, dslam_hash_entry *o) {
  memcpy(o, s->hash.data(), s->hash.size() * sizeof(dslam_hash_entry)); return 0;
}
extern "C" int oracle_download_voxel_blocks(oracle_engine *, const oracle_scene *s, int first, int n, dslam_voxel *o) {
  memcpy(o, &s->vba[(size_t)first * 512], (size_t)n * 512 * sizeof(dslam_voxel)); return 0;
}
extern "C" int oracle_download_allocation_list(oracle_engine *, const oracle_scene *s, int32_t *o) {
  memcpy(o, s->alloc_list.data(), s->alloc_list.size() * 4); return 0;
}
extern "C" int oracle_download_excess_list(oracle_engine *, const oracle_scene *s, int32_t *o) {
  memcpy(o, s->excess_list.data(), s->excess_list.size() * 4); return 0;
}
extern "C" int oracle_download_visible_ids(oracle_engine *, const oracle_render_state *r, int32_t *o, int cap, int *count) {
  int n = std::min(cap, r->no_visible);
  memcpy(o, r->visible_ids.data(), (size_t)n * 4);
  if (count) *count = r->no_visible;
  return 0;
}
extern "C" int oracle_download_visible_types(oracle_engine *, const oracle_render_state *r, uint8_t *o) {
  memcpy(o, r->visible_type.data(), r->visible_type.size()); return 0;
}
extern "C" int oracle_download_range_image(oracle_engine *, const oracle_render_state *r, float *o) {
  memcpy(o, r->range.data(), r->range.size() * sizeof(V2f)); return 0;
}
extern "C" int oracle_download_raycast_result(oracle_engine *, const oracle_render_state *r, float *o) {
  memcpy(o, r->raycast.data(), r->raycast.size() * sizeof(V4f)); return 0;
}
extern "C" int oracle_download_view_depth(oracle_engine *, const oracle_view *v, float *o) {
  memcpy(o, v->depth.data(), v->depth.size() * 4); return 0;
}
extern "C" int oracle_download_swap_states(oracle_engine *, const oracle_scene *s, uint8_t *o) {
  if (s->swap_state.empty()) return DSLAM_ERR_INVALID;
  memcpy(o, s->swap_state.data(), s->swap_state.size()); return 0;
}
extern "C" int oracle_download_alloc_scratch(oracle_engine *, const oracle_scene *s, uint8_t *types, int16_t *coords) {
  if (types) memcpy(types, s->alloc_type.data(), s->alloc_type.size());
  if (coords) memcpy(coords, s->block_coords.data(), s->block_coords.size() * sizeof(S4));
  return 0;
}
extern "C" int oracle_download_last_seen(oracle_engine *, const oracle_scene *s, int32_t *o) {
  memcpy(o, s->last_seen.data(), s->last_seen.size() * 4); return 0;
}
// host store of one entry (ITMGlobalCache::GetStoredVoxelBlock); returns has_stored
extern "C" int oracle_download_stored_block(oracle_engine *, const oracle_scene *s, int entry, dslam_voxel *o) {
  if (s->swap_state.empty()) return DSLAM_ERR_INVALID;
  if (o) memcpy(o, s->stored + (size_t)entry * 512, 512 * sizeof(dslam_voxel));
  return s->has_stored[entry] ? 1 : 0;
}
extern "C" int oracle_upload_scene_state(oracle_engine *, oracle_scene *s, const dslam_hash_entry *hash, const int32_t *alloc_list,
                              int last_free, const int32_t *excess_list, int last_free_ex) {
  if (hash) memcpy(s->hash.data(), hash, s->hash.size() * sizeof(dslam_hash_entry));
  if (alloc_list) { memcpy(s->alloc_list.data(), alloc_list, s->alloc_list.size() * 4); s->last_free = last_free; }
  if (excess_list) { memcpy(s->excess_list.data(), excess_list, s->excess_list.size() * 4); s->last_free_ex = last_free_ex; }
  return 0;
}
extern "C" int oracle_upload_voxel_blocks(oracle_engine *, oracle_scene *s, int first, int n, const dslam_voxel *h) {
  memcpy(&s->vba[(size_t)first * 512], h, (size_t)n * 512 * sizeof(dslam_voxel)); return 0;
}
extern "C" int oracle_upload_visible_ids(oracle_engine *, oracle_render_state *r, const int32_t *ids, int count) {
  memcpy(r->visible_ids.data(), ids, (size_t)count * 4); r->no_visible = count; return 0;
}

extern "C" int oracle_raycast_debug_buffer(int *buf) { g_dbg_pixel_steps = buf; return 0; }
// analysis aid: per ray and march step a kind byte -- 1 no block, 2 plain (block cached), 3 plain after a probe,
// 4/5 near-surface cell straddling blocks (new / same 2x2x2 neighbourhood as the ray's previous such step), 6/7 the
// same after a probe; buf = NULL switches it off
extern "C" int oracle_raycast_trace_buffer(uint8_t *buf, int steps_per_ray) { g_dbg_trace = buf; g_dbg_trace_len = steps_per_ray; return 0; }
extern "C" int oracle_raycast_stats(long long *out4, int reset) {
  out4[0] = g_dbg_steps; out4[1] = g_dbg_interp; out4[2] = g_dbg_rays; out4[3] = g_dbg_maxsteps;
  if (reset) { g_dbg_steps = g_dbg_interp = g_dbg_rays = g_dbg_maxsteps = 0; }
  return 0;
}

// known-answer helpers
extern "C" int oracle_hash_index(int bx, int by, int bz, int num_buckets) { return hash_index(bx, by, bz, (uint32_t)(num_buckets - 1)); }
extern "C" int oracle_point_to_block(int px, int py, int pz, int *b) {
  V3i p = {px, py, pz}, bb;
  int lin = point_to_block(p, bb);
  b[0] = bb.x; b[1] = bb.y; b[2] = bb.z;
  return lin;
}
extern "C" int oracle_invert_matrix(const float *m, float *out) { return inv4(m, out) ? 0 : 1; }
// one voxel, depth part only: eta sequence -> (sdf, w) (Appendix C "integrate one voxel")
extern "C" int oracle_update_voxel_eta(int16_t *sdf, uint8_t *w, float eta, float mu, int maxW) {
  float oldF = sdf_to_float(*sdf);
  int oldW = *w;
  float newF = std::min(1.0f, eta / mu);
  int newW = 1;
  newF = (float)oldW * oldF + (float)newW * newF;
  newW = oldW + newW;
  newF /= (float)newW;
  newW = std::min(newW, maxW);
  *sdf = float_to_sdf(newF);
  *w = (uint8_t)newW;
  return 0;
}

