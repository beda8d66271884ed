// dslam_internal.h -- host-side objects behind the opaque handles of include/dslam_fusion.h.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dslam_fusion.h"
#include "dslam_device.h"

namespace dslam {

void set_last_error(const std::string &msg);
int hip_fail(hipError_t err, const char *what, const char *file, int line);

#define DSLAM_HIP(call)                                                       \
  do {                                                                        \
    hipError_t _e = (call);                                                   \
    if (_e != hipSuccess) return ::dslam::hip_fail(_e, #call, __FILE__, __LINE__); \
  } while (0)

#define DSLAM_REQUIRE(cond, msg)                 \
  do {                                           \
    if (!(cond)) {                               \
      ::dslam::set_last_error(msg);              \
      return DSLAM_ERR_INVALID;                  \
    }                                            \
  } while (0)

// host inverse of a column-major 4x4 (ORUtils::Matrix4::inv); used for invM_d = pose_d->GetInvM()
bool invert_matrix(const float *m, float *dst);

}  // namespace dslam

struct dslam_engine {
  int device = 0;
  hipStream_t stream = nullptr;
  // A caller's fence recorded behind the last call that read a view is also that view's "consumed" mark: the pipelined
  // upload then records no event of its own (an event record costs the compute stream ~3.5 us per frame, measured).
  unsigned long long view_reads = 0;   // calls that enqueued kernels reading a view's images, so far
  dslam_fence *last_fence = nullptr;   // most recently recorded fence
  std::vector<dslam_fence *> fences;   // every fence of this engine that still exists: live ones, and destroyed ones whose
                                       // event a view still waits on (they go when the last such view lets go)
  hipStream_t copy_stream = nullptr;  // pipelined uploads (async mode, page-locked sources): H2D of frame i + 1 under frame i's kernels
  bool async_mode = false;
  dslam_weight_params wp{0, 1, 1.0f};
  // scratch shared by all scenes of this engine (sized for the largest scene seen)
  int scratch_entries = 0;
  int scratch_local_blocks = 0;
  unsigned *order_keys = nullptr;     // [entries] mark-phase order keys; ALL ZERO between allocation passes (the pass that
                                      // sets a key clears it again, so no pass starts with a 4.7 MB memset)
  unsigned char *alloc_type = nullptr;  // [entries] entriesAllocType of the last pass (kept for the parity tests)
  short4 *block_coords = nullptr;     // [entries] blockCoords
  // bit-packed summaries of an allocation pass (dslam_device.h): requests on empty bucket heads (q1) / chain ends (q2),
  // entries a pixel's walk found (mark).  Two sets used alternately: pass k sets bits in set k & 1 and zeroes set
  // (k + 1) & 1 -- the set of the pass before it -- so no pass starts with a memset; bits_dirty[s] = words of set s that
  // may be non-zero (scenes of different sizes share the sets).  retest: outcome of the frustum re-test of the entries
  // that were visible before the pass (every word is written by every pass).
  unsigned *bits_q1[2] = {nullptr, nullptr}, *bits_q2[2] = {nullptr, nullptr}, *bits_mark[2] = {nullptr, nullptr};
  unsigned *bits_retest = nullptr;
  int bits_words = 0;                 // words per bitmap (whole tiles)
  int bits_dirty[2] = {0, 0};
  unsigned alloc_pass = 0;
  // tickets of the single-pass ordered compactions (dslam_device.h take_ticket): one ever-growing device counter and
  // the value the host knows it has
  unsigned *ticket = nullptr;         // device [16]: counter k is ticket + k
  unsigned ticket_base = 0;           // counter 0
  unsigned ticket_base2 = 0;          // counter 1 (a second, independent chain inside the same launch)
  unsigned long long hip_failures_seen = 0;  // ... which is only as good as the launches that were accepted: after any HIP
                                      // failure of the process the bases are read back from the device (tickets_resync)
  // what kernels could not tell anybody (report_error, dslam_device.h): one page-locked word, looked at by every entry
  // point that waits for the stream (sync_check).  Sticky per scene in SceneCounters::error_flags; here: told once.
  int *err_host = nullptr;
  // single-pass ordered compactions: per-tile aggregates published inside one launch ({epoch, counts} in one 8-byte
  // word per tile; three channels: allocation requests, commit results, visible counts) and the launch counter that
  // tags them, so the arrays never need clearing
  unsigned long long *agg = nullptr;  // [3][agg_tiles]
  int agg_tiles = 0;
  unsigned epoch = 0;
  int sweep_grid_cap = 0;             // workgroups of the sweep kernels that are certainly co-resident on this device
  int *list_d = nullptr;              // [max(local_blocks, entries)] general purpose scratch (bucket leaders, live flags)
  // release pipeline (decay / sliding window): flags that are ALL ZERO between passes (the kernels that consume a flag
  // clear it), so no pass starts with a memset
  unsigned char *rem_flags = nullptr;    // [entries] entry is being released
  unsigned char *freed_flags = nullptr;  // [entries] excess slot came free
  unsigned char *rem_cand = nullptr;     // [local_blocks] candidate i of the decay pass lost its last measured voxel
  int *maint_flags = nullptr;            // device [4]: [0] the pass took something out of a visible list
  int *tile_counts = nullptr;         // [2 * tiles] per-tile counts of the ordered compactions
  int *tile_offsets = nullptr;        // [2 * tiles]
  int *list_a = nullptr;              // [max(local_blocks, entries)] general purpose int lists
  int *list_b = nullptr;
  int *list_c = nullptr;
  short4 *pos_scratch = nullptr;      // [local_blocks]
  // pinned staging
  void *pinned = nullptr;             // small host mirror for counters / stats
  size_t pinned_bytes = 0;
  void *staging_dev = nullptr;        // H2D staging for view uploads (rgba + depth)
  size_t staging_bytes = 0;
  void *staging_host = nullptr;
  // kernel timer (bench roofline): HIP events around the integrate kernel on the engine stream
  bool timer_enabled = false;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  double timer_ms = 0;
  long long timer_launches = 0;
  long long timer_blocks = 0;
  int *timer_counts_dev = nullptr;    // visible-block count of each timed launch (written by the kernel)
  int sm_count = 256;
  // visible blocks from which on a fusion launch counts as larger than the Infinity Cache (65536 x 4 KiB = 256 MiB): trailing
  // push workgroups (integrate.hip kPushJobMin, decided on the device) and streaming cache policy (decided by the host from
  // dslam_render_state::vis_hint).  Lowered only by the parity test of those paths.
  int push_job_min = 65536;
  long long stream_launches = 0;      // fusion / de-integration launches that took the streaming instantiation (test hook)
  int render_tile_budget = DSLAM_MAX_RENDERING_BLOCKS;  // MAX_RENDERING_BLOCKS; lowered only by the budget test
  double *icp_partials_host = nullptr;  // depth tracker: per-workgroup partial sums in mapped pinned host memory
  double *icp_partials = nullptr;       // ... and the device address of the same buffer
  int *misc_counter = nullptr;        // device: small result counters of one-off kernels (depthPostProcessing)
  // the last mesh dslam_mesh_scene produced (ITMMesh: triangles as 3 x Vector3f, metres)
  float *mesh_positions = nullptr;    // device [mesh_triangles][3][3]
  float *mesh_colours = nullptr;      // device [mesh_triangles][3][3], only if asked for
  size_t mesh_bytes = 0;              // capacity of each of the two buffers
  int mesh_triangles = 0;
  bool mesh_has_colour = false;
  bool mesh_table_ready = false;      // the case table sits in this device's constant memory
};

struct dslam_scene {
  dslam_engine *engine = nullptr;
  dslam_scene_params p{};
  int n_entries = 0;
  dslam::HashEntry *hash = nullptr;
  uint2 *voxels = nullptr;
  bool voxels_external = false;
  int *alloc_list = nullptr;
  int *excess_list = nullptr;
  int *last_seen = nullptr;           // per voxel-block slot
  dslam::SceneCounters *counters = nullptr;  // device
  // visible-list history: per voxel-block slot two bit rings (0 fusion, 1 defusion); list k of ring q
  // owns bit k % (64*history_words) of masks[(slot*2+q)*history_words ...]
  int history_words = 4;
  unsigned long long *masks = nullptr;  // device
  int ring_head[2] = {0, 0}, ring_next[2] = {0, 0}, decay_cursor[2] = {0, 0};
  int frame_counter = 0;
  // ITMGlobalCache
  unsigned char *swap_state = nullptr;  // device [entries]
  unsigned *alloc_bits = nullptr;       // device [bit tiles]: bit t = entry t holds a resident block (ptr >= 0)
  unsigned *swap1_bits = nullptr;       // device [bit tiles], swapping only: bit t = swap_state[t] == 1 (host copy to be merged)
  // host store of swapped-out blocks: page-locked slabs the kernels read and write directly over PCIe (no staging
  // copy, no host memcpy); an entry's block lives in slot slot_dev[entry]; slots are handed out by an atomic counter
  std::vector<uint4 *> slabs;           // pinned, kSlabBlocks blocks each, allocated as the store grows
  uint4 **slab_ptrs_dev = nullptr;      // device [kMaxSlabs]: the same pointers for the kernels
  int *slot_dev = nullptr;              // device [entries]: slot of the entry's stored block, -1 = none.  The table and
                                        // the slot counter (SceneCounters::next_slot) live on the device: a swap batch
                                        // needs no host round trip (round 1: two per ProcessFrame)
  int *next_slot_host = nullptr;        // pinned: copy of next_slot queued behind every batch that may hand out slots
  long long slot_bound = 0;             // upper bound of next_slot the host can prove (slabs exist for all of it)
  int shard = 0, num_shards = 1, chunk_blocks = 256;
  unsigned long long version = 0;       // bumped by every call that can change the map (GetImage memo key)
  int shard_first = 0, shard_count = -1;  // contiguous slot range (count < 0: off)
  // sharded re-integration: per voxel-block slot "a (de-)integration pass visited this block since tracking began"
  unsigned char *dirty = nullptr;       // device [num_local_blocks], allocated by dslam_scene_track_dirty
  bool dirty_tracking = false;
  int *dirty_list = nullptr;            // device [num_local_blocks]: the dirty slots in virtual (shard-major) order
  int *dirty_counts = nullptr;          // device [128]: per shard its number of dirty slots (+ scratch)
  // block-major re-integration batch (dslam_reintegrate_batch): per voxel-block slot the re-fusion pass of the batch that
  // allocated it (0: it existed before), which operations of the batch touch it, an entry that holds it; the list of
  // touched slots; [0] its length, [1] the work cursor.  alloc_born / alloc_born_stamp: the allocation sweep stamps the
  // blocks it commits while a batch is being planned
  int *batch_born = nullptr;
  unsigned long long *batch_opmask = nullptr;
  int *batch_slot_entry = nullptr, *batch_order = nullptr, *batch_counters = nullptr;   // (batch_order: 8 class lists)
  unsigned char *batch_marks = nullptr;   // [num_local_blocks][64]: operation k of the batch touches the block (zero between batches)
  void *batch_ops_dev = nullptr, *batch_lists_dev = nullptr;
  float *batch_depth = nullptr;           // one float depth image per keyframe of a batch (written by its allocation pass, read by
  size_t batch_depth_pixels = 0;          // both of its operations in the block launch); pixels per image it was sized for
  void *batch_staging = nullptr;          // page-locked: the operations and list references of a batch on their way to the device
  hipEvent_t batch_staging_ev = nullptr;  // ... the copies out of it have been made
  int *alloc_born = nullptr;
  int alloc_born_stamp = 0;
  int dirty_shards = 0, dirty_chunk = 0;  // the layout of the last dslam_shard_dirty_plan
};

struct dslam_render_state {
  dslam_engine *engine = nullptr;
  int w = 0, h = 0, n_entries = 0, n_local = 0;
  int *visible_ids = nullptr;
  unsigned char *visible_type = nullptr;
  unsigned *vis_bits = nullptr;   // bit t = visible_type[t] != 0 (kept by every kernel that writes a type)
  float2 *range = nullptr;      // renderingRangeImage (full image stride)
  float4 *raycast = nullptr;    // raycastResult
  uchar4 *image_rgba = nullptr; // RenderImage's outputImage (rgba types)
  float *image_float = nullptr;
  float4 *icp_points = nullptr, *icp_normals = nullptr;  // allocated on first use
  uchar4 *raycast_image = nullptr;  // ITMRenderState::raycastImage: the grey tracking raycast CreateICPMaps draws (with the maps)
  int4 *proj_boxes = nullptr;   // per visible block: render bbox (ul.x, ul.y, lr.x, lr.y)
  float2 *proj_z = nullptr;     // per visible block: z range
  int *proj_req = nullptr;      // per visible block: render tiles required (0 = invalid projection)
  int *proj_wg_tiles = nullptr; // render tiles requested per workgroup of the projection pass (summed by the next kernel)
  dslam::RenderCounters *counters = nullptr;  // device
  // One page-locked word: the length of the visible list as an allocation pass that RAN left it -- k_alloc_sweep rewrites it
  // whenever the answer to "at least push_job_min blocks?" would change, an uploaded list sets it.  The
  // host looks at it -- without waiting for anything, so on an asynchronous engine it is a frame or two late -- to choose the
  // fusion kernel's cache policy (launch_integrate): a hint, both policies compute the same bytes.
  int *vis_hint = nullptr;
  // entriesVisibleType carries a generation bit (0x80): an allocation pass writes its marks with the pass' bit, so a
  // non-zero byte with the OTHER bit is "visible in the previous pass" (upstream's re-arming of the previous visible
  // list as type 3) without a pass over that list.  The C ABI hands out the plain types (bit masked off).
  unsigned char gen = 0;
  // the visible list was replaced behind the types' back (FindVisibleBlocks on this render state, an uploaded list):
  // the next allocation pass re-derives the "previously visible" marks from the list first
  bool types_follow_list = true;
  // GetImage memo: raycastResult (and the visible list / range image behind it) is still that of this scene version,
  // pose and intrinsics, so another image type of the same view only has to be shaded (the reference's GUI asks for
  // a depth and a colour image of the same free pose every tick, DenseSlam.h:146-164)
  bool memo_valid = false;
  const dslam_scene *memo_scene = nullptr;
  unsigned long long memo_version = 0;
  int memo_budget = 0;
  float memo_M[16] = {0}, memo_intr[4] = {0};
};

struct dslam_view {
  dslam_engine *engine = nullptr;
  int w_rgb = 0, h_rgb = 0, w_d = 0, h_d = 0;
  uchar4 *rgba = nullptr;       // own buffers (host uploads land here)
  float *depth = nullptr;
  short *raw_depth = nullptr;
  mutable float *pyramid = nullptr;  // depth tracker: levels 1.. of the depth pyramid (scratch, allocated on first use)
  // what the kernels read: own buffers, or the caller's resident frame (dslam_view_update_device: no copy)
  const uchar4 *rgba_src = nullptr;
  const short *raw_src = nullptr;
  // pipelined uploads (async engine + page-locked caller images): two landing buffers, filled on the engine's copy
  // stream while the compute stream still reads the other one
  uchar4 *up_rgba[2] = {nullptr, nullptr};
  short *up_raw[2] = {nullptr, nullptr};
  hipEvent_t up_done[2] = {nullptr, nullptr};      // copy stream: buffer b has landed
  hipEvent_t up_consumed[2] = {nullptr, nullptr};  // compute stream: every kernel that reads buffer b has been passed
  hipEvent_t up_consumed_by[2] = {nullptr, nullptr};  // the event that says so for the current contents: up_consumed[b] or a caller's fence
  dslam_fence *up_lender[2] = {nullptr, nullptr};     // ... the fence that event belongs to, if it is a caller's
  bool up_used[2] = {false, false};
  int up_next = 0;
  float affine_a = 0.001f, affine_b = 0.0f;
  mutable bool depth_dirty = false;  // float depth not yet derived from raw_src (done by the next consumer)
  double timestamp = 0;
};

// a marker in the engine's stream (dslam_fence_*): lets a pipelining caller learn when a frame's results have landed
struct dslam_fence {
  dslam_engine *engine = nullptr;
  hipEvent_t ev = nullptr;
  bool recorded = false;
  unsigned long long view_reads_at_record = 0;  // engine->view_reads when it was recorded
  int lent_count = 0;   // views that wait on `ev` as their "landing buffer consumed" mark at the moment
  bool zombie = false;  // destroyed by the caller while lent: the object and its event live until the last view lets go
};

// mfusionFrameDataBase's image payload (fusionFrameInfo::rgbinfo / depthinfo, DenseSlam.h:431-433) kept in HBM:
// `capacity` slots of one RGBA image + one int16 depth image each, in two contiguous arrays
struct dslam_frame_store {
  dslam_engine *engine = nullptr;
  int w_rgb = 0, h_rgb = 0, w_d = 0, h_d = 0, capacity = 0;
  size_t rgba_bytes = 0, depth_bytes = 0;  // per slot
  unsigned char *rgba = nullptr;
  unsigned char *depth = nullptr;
  // optional: per slot the visible list of the keyframe's fusion (dslam_frame_store_enable_lists):
  // [RenderCounters-sized header: count][int ids[list_cap]][short4 pos[list_cap]]
  unsigned char *lists = nullptr;
  int list_cap = 0;
  int list_entries = 0;   // hash-table size (entries) of the scene the lists were enabled for: their ids index that table
  size_t list_bytes = 0;  // per slot
  std::vector<unsigned char> has_list;
  // where each slot's list lives: a buffer of `lists`, or -- after a re-integration batch, which writes the lists of its
  // re-fusions to scratch buffers and then trades buffers instead of copying -- one of `batch_lists`
  std::vector<unsigned char *> list_ptr;
  unsigned char *batch_lists = nullptr;          // 32 more list buffers (allocated by the first batch)
  std::vector<unsigned char *> batch_list_ptr;
};

namespace dslam {
// diagnostics (DSLAM_DEBUG_SYNC=1): wait for the stream after a launch and say which one it was -- finds the kernel that
// hangs or faults without a profiler
inline void dbg_sync(dslam_engine *e, const char *what) {
  static const bool on = getenv("DSLAM_DEBUG_SYNC") != nullptr;
  if (!on) return;
  fprintf(stderr, "[dslam] %s ...", what);
  fflush(stderr);
  const hipError_t err = hipStreamSynchronize(e->stream);
  fprintf(stderr, " %s\n", err == hipSuccess ? "ok" : hipGetErrorString(err));
  fflush(stderr);
}
// kernels' host launchers (one translation unit per subsystem)
int launch_scene_reset(dslam_engine *e, dslam_scene *s);
int launch_inject_error(dslam_engine *e, dslam_scene *s, int bits);
int launch_build_alloc_bits(dslam_engine *e, dslam_scene *s);  // alloc_bits from an uploaded table
int launch_view_convert(dslam_engine *e, dslam_view *v, const void *rgba_dev, const void *depth_dev, float a, float b);
int launch_bgr_to_rgba(dslam_engine *e, const void *bgr_dev, uchar4 *rgba_dev, int npix);
int launch_bilateral(dslam_engine *e, dslam_view *v);
int launch_dataset_depth(dslam_engine *e, short *depth_dev, int n, int format, float max_depth_m);
int launch_depth_to_int16(dslam_engine *e, const float *depth_dev, short *out_dev, int n, int scale);
int launch_depth_post(dslam_engine *e, short *curr_dev, const unsigned short *prev_dev, int cols, int rows,
                      const float *Tpc, const float *intr, float threshold, float area, int *count_dev);
// list_out / count_out: write the pass' visible list there instead of into the render state's own list (whose bitmap and
// types follow the pass either way; the re-integration batch keeps the lists of all its passes)
int launch_allocate(dslam_engine *e, dslam_scene *s, const dslam_view *v, dslam_render_state *r, const float *M_d,
                    const float *intr, int only_update_visible_list, int *list_out = nullptr, void *count_out = nullptr);
// push_ring >= 0: also queue the frame's visible list on that ring (fused into the integrate kernel)
int launch_integrate(dslam_engine *e, dslam_scene *s, const dslam_view *v, const dslam_render_state *r,
                     const float *M_d, const float *intr_d, const float *M_rgb, const float *intr_rgb,
                     bool deintegrate, int push_ring = -1);
// the same kernel over a stored list (count header, ids, expected block positions) instead of a render state's
int launch_integrate_list(dslam_engine *e, dslam_scene *s, const dslam_view *v, const void *count_header, const int *ids,
                          const short4 *expect_pos, const float *M_d, const float *intr_d, const float *M_rgb,
                          const float *intr_rgb, bool deintegrate);
int launch_store_visible_list(dslam_engine *e, const dslam_scene *s, const dslam_render_state *r, void *header, int *ids,
                              short4 *pos, int capacity);
int launch_store_list_positions(dslam_engine *e, const dslam_scene *s, const void *jobs_dev, int n_jobs);
int launch_batch_ops(dslam_engine *e, const void *lists_dev, int n_ops, const dslam_scene *s, const int *born, unsigned char *marks,
                     unsigned long long *opmask, int *slot_entry, int *cls_list, int *cls_count);
int launch_reintegrate_blocks(dslam_engine *e, dslam_scene *s, int w_d, int h_d, int w_rgb, int h_rgb, const float *intr,
                              const void *ops_dev, const unsigned long long *opmask, const int *slot_entry,
                              const int *cls_list, const int *cls_count, int push_ring, int n_ops);
int alloc_step_cap(const dslam_scene *s, int W, int H, int *cap_out);
int ensure_view_depth(dslam_engine *e, const dslam_view *v);
int launch_selftest_division(dslam_engine *e, long long samples, unsigned long long *mismatches_dev);
int prepare_push_visible_list(dslam_engine *e, dslam_scene *s, int q, int *bit_out, int *frame_out);
int launch_find_visible(dslam_engine *e, const dslam_scene *s, dslam_render_state *r, const float *M,
                        const float *intr);
int launch_count_visible(dslam_engine *e, const dslam_scene *s, const dslam_render_state *r, int min_id, int max_id,
                         int *out);
int launch_expected_depths(dslam_engine *e, const dslam_scene *s, dslam_render_state *r, const float *M,
                           const float *intr);
int launch_find_visible_and_depths(dslam_engine *e, const dslam_scene *s, dslam_render_state *r, const float *M,
                                   const float *intr);
int launch_track_camera(dslam_engine *e, const dslam_view *v, dslam_render_state *r, const float *scenePose, float *pose_M,
                        const float *intr, const dslam_tracker_params *tp, dslam_tracker_result *res);
int launch_render(dslam_engine *e, const dslam_scene *s, dslam_render_state *r, const float *M, const float *intr,
                  int type, bool reuse_raycast = false, void *image_out_override = nullptr);
int launch_icp_maps(dslam_engine *e, const dslam_scene *s, dslam_render_state *r, const float *M, const float *intr);
int launch_mesh_scene(dslam_engine *e, const dslam_scene *s, int max_triangles, int with_colour, int *out_num);
int launch_decay(dslam_engine *e, dslam_scene *s, dslam_render_state *r, int max_weight, int min_age, int force_all,
                 int which);
int launch_slide_pop(dslam_engine *e, dslam_scene *s, dslam_render_state *r, int which);
int launch_swap_in(dslam_engine *e, dslam_scene *s, dslam_render_state *r);
int launch_swap_out(dslam_engine *e, dslam_scene *s, dslam_render_state *r, bool ignore_visibility);
int launch_save_to_global(dslam_engine *e, dslam_scene *s);
int launch_dirty_plan(dslam_engine *e, dslam_scene *s, int num_shards, int chunk_blocks, int *counts_host);
int launch_dirty_pack(dslam_engine *e, const dslam_scene *s, int shard, void *send_dev, int capacity_blocks);
int launch_dirty_unpack(dslam_engine *e, dslam_scene *s, int skip_shard, const void *recv_dev, int stride_blocks);
int ensure_scratch(dslam_engine *e, int entries, int local_blocks);
int finish_call(dslam_engine *e);  // synchronise unless the engine is in async mode
int device_errors(dslam_engine *e);  // report (once) what kernels left in dslam_engine::err_host
int sync_check(dslam_engine *e);   // wait for the engine's stream, then report what kernels left in dslam_engine::err_host
// The host advances its copy of the ticket counters when it enqueues a launch.  A launch the runtime rejected never moves the
// device counters, and every later ticket would then lie in front of its base (the kernels compare unsigned, so such a
// launch does nothing instead of indexing with a negative tile -- but it does nothing).  Any HIP failure bumps a process-
// wide count (hip_fail); an engine that sees a new value drains its stream and reads the counters back.
unsigned long long hip_failure_count();
int tickets_resync(dslam_engine *e);
inline int tickets_ok(dslam_engine *e) { return e->hip_failures_seen == hip_failure_count() ? DSLAM_OK : tickets_resync(e); }
inline int num_tiles(int entries) { return (entries + kTileEntries - 1) / kTileEntries; }
}  // namespace dslam
