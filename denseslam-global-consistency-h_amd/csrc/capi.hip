// capi.hip -- the C ABI of include/dslam_fusion.h: object lifetime, host<->device plumbing, and the
// composition of kernels into the ITMLib engine calls (ITMDenseMapper::ProcessFrame, ITMMainEngine::GetImage ...).
// No arithmetic of the hot path lives here; there is no CPU fallback: without a HIP device the engine cannot be
// created and every entry point fails.
#include <cctype>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

#include "dslam_internal.h"

#pragma clang fp contract(off)

namespace dslam {

static thread_local std::string g_last_error;
void set_last_error(const std::string &msg) { g_last_error = msg; }
static std::atomic<unsigned long long> g_hip_failures{0};
unsigned long long hip_failure_count() { return g_hip_failures.load(std::memory_order_relaxed); }
int hip_fail(hipError_t err, const char *what, const char *file, int line) {
  g_hip_failures.fetch_add(1, std::memory_order_relaxed);
  char buf[512];
  snprintf(buf, sizeof(buf), "%s failed: %s (%s:%d)", what, hipGetErrorString(err), file, line);
  g_last_error = buf;
  return DSLAM_ERR_HIP;
}

// ORUtils::Matrix4::inv restated (cofactor expansion); identical expression order to the CPU oracle so both
// derive bit-identical inverse poses.  Host code: compiled with -ffp-contract=off.
bool invert_matrix(const float *m, float *dst) {
  float tmp[12], src[16], det;
  for (int i = 0; i < 4; i++) {
    src[i] = m[i * 4];
    src[i + 4] = m[i * 4 + 1];
    src[i + 8] = m[i * 4 + 2];
    src[i + 12] = m[i * 4 + 3];
  }
  tmp[0] = src[10] * src[15]; tmp[1] = src[11] * src[14]; tmp[2] = src[9] * src[15]; tmp[3] = src[11] * src[13];
  tmp[4] = src[9] * src[14]; tmp[5] = src[10] * src[13]; tmp[6] = src[8] * src[15]; tmp[7] = src[11] * src[12];
  tmp[8] = src[8] * src[14]; tmp[9] = src[10] * src[12]; tmp[10] = src[8] * src[13]; tmp[11] = src[9] * src[12];
  dst[0] = (tmp[0] * src[5] + tmp[3] * src[6] + tmp[4] * src[7]) - (tmp[1] * src[5] + tmp[2] * src[6] + tmp[5] * src[7]);
  dst[1] = (tmp[1] * src[4] + tmp[6] * src[6] + tmp[9] * src[7]) - (tmp[0] * src[4] + tmp[7] * src[6] + tmp[8] * src[7]);
  dst[2] = (tmp[2] * src[4] + tmp[7] * src[5] + tmp[10] * src[7]) - (tmp[3] * src[4] + tmp[6] * src[5] + tmp[11] * src[7]);
  dst[3] = (tmp[5] * src[4] + tmp[8] * src[5] + tmp[11] * src[6]) - (tmp[4] * src[4] + tmp[9] * src[5] + tmp[10] * src[6]);
  dst[4] = (tmp[1] * src[1] + tmp[2] * src[2] + tmp[5] * src[3]) - (tmp[0] * src[1] + tmp[3] * src[2] + tmp[4] * src[3]);
  dst[5] = (tmp[0] * src[0] + tmp[7] * src[2] + tmp[8] * src[3]) - (tmp[1] * src[0] + tmp[6] * src[2] + tmp[9] * src[3]);
  dst[6] = (tmp[3] * src[0] + tmp[6] * src[1] + tmp[11] * src[3]) - (tmp[2] * src[0] + tmp[7] * src[1] + tmp[10] * src[3]);
  dst[7] = (tmp[4] * src[0] + tmp[9] * src[1] + tmp[10] * src[2]) - (tmp[5] * src[0] + tmp[8] * src[1] + tmp[11] * src[2]);
  tmp[0] = src[2] * src[7]; tmp[1] = src[3] * src[6]; tmp[2] = src[1] * src[7]; tmp[3] = src[3] * src[5];
  tmp[4] = src[1] * src[6]; tmp[5] = src[2] * src[5]; tmp[6] = src[0] * src[7]; tmp[7] = src[3] * src[4];
  tmp[8] = src[0] * src[6]; tmp[9] = src[2] * src[4]; tmp[10] = src[0] * src[5]; tmp[11] = src[1] * src[4];
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

// Map versions are drawn from one process-wide counter, so a version never repeats -- not even on a new scene that
// happens to be allocated where a destroyed one was (the GetImage memo compares scene pointer and version).
unsigned long long next_map_version() {
  static std::atomic<unsigned long long> counter{0};
  return ++counter;
}

// Every entry point that waits for the stream ends here: what a kernel reported through report_error (a tile count that
// never arrived, an allocation ray longer than the order key encodes) is returned by the first call that could know --
// the call itself on a synchronous engine, the next synchronising call (fence wait, read-back, dslam_engine_synchronize)
// on an asynchronous one.  Told once per occurrence; the scene's own flags stay set for dslam_get_stats.
int sync_check(dslam_engine *e) {
  DSLAM_HIP(hipStreamSynchronize(e->stream));
  return device_errors(e);
}
int device_errors(dslam_engine *e) {
  int *h = e->err_host;
  if (!h || __atomic_load_n(h, __ATOMIC_RELAXED) == 0) return DSLAM_OK;
  const int flags = __atomic_exchange_n(h, 0, __ATOMIC_RELAXED);   // (taken in one step: a report that lands now is the next call's)
  if (flags == 0) return DSLAM_OK;
  if (flags & 2) {
    set_last_error("a tile count of an ordered compaction never arrived (device made no progress?): the map state is undefined");
    return DSLAM_ERR_HIP;
  }
  set_last_error("allocation ray needed more steps than the order key encodes (non-rigid pose or mu/voxel_size changed?)");
  return DSLAM_ERR_UNSUPPORTED;
}

int finish_call(dslam_engine *e) {
  if (!e->async_mode) return sync_check(e);
  return DSLAM_OK;
}

int tickets_resync(dslam_engine *e) {
  const unsigned long long seen = hip_failure_count();
  unsigned host[2] = {0, 0};
  if (hipStreamSynchronize(e->stream) != hipSuccess || hipMemcpy(host, e->ticket, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) {
    (void)hipGetLastError();
    set_last_error("ticket counters could not be read back after a HIP failure");
    return DSLAM_ERR_HIP;   // (hip_failures_seen stays behind: the next pass tries again)
  }
  e->ticket_base = host[0];
  e->ticket_base2 = host[1];
  e->hip_failures_seen = seen;
  return DSLAM_OK;
}

template <typename T>
static void free_dev(T *&p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}

int ensure_scratch(dslam_engine *e, int entries, int local_blocks) {
  if (entries <= e->scratch_entries && local_blocks <= e->scratch_local_blocks) return DSLAM_OK;
  DSLAM_HIP(hipStreamSynchronize(e->stream));
  const int N = entries > e->scratch_entries ? entries : e->scratch_entries;
  const int L = local_blocks > e->scratch_local_blocks ? local_blocks : e->scratch_local_blocks;
  free_dev(e->order_keys); free_dev(e->alloc_type); free_dev(e->block_coords); free_dev(e->tile_counts); free_dev(e->tile_offsets);
  free_dev(e->list_a); free_dev(e->list_b); free_dev(e->list_c); free_dev(e->list_d); free_dev(e->pos_scratch);
  free_dev(e->agg);
  for (int k = 0; k < 2; k++) { free_dev(e->bits_q1[k]); free_dev(e->bits_q2[k]); free_dev(e->bits_mark[k]); }
  free_dev(e->bits_retest);
  free_dev(e->rem_flags); free_dev(e->freed_flags); free_dev(e->rem_cand); free_dev(e->maint_flags);
  // order keys and allocType: cleared here once, kept clean by the allocation passes (scenes of different sizes share
  // them, so both start at fixed addresses: a pass only ever touches [0, its entry count) of each)
  DSLAM_HIP(hipMalloc(&e->order_keys, (size_t)N * 4));
  DSLAM_HIP(hipMemsetAsync(e->order_keys, 0, (size_t)N * 4, e->stream));
  DSLAM_HIP(hipMalloc(&e->alloc_type, (size_t)N));
  DSLAM_HIP(hipMemsetAsync(e->alloc_type, 0, (size_t)N, e->stream));
  DSLAM_HIP(hipMalloc(&e->block_coords, (size_t)N * sizeof(short4)));
  // the bitmaps of the allocation pass (whole tiles; the alternating sets start clean and are kept clean by the passes)
  e->bits_words = bit_tiles(N) * kBitTileWords;
  const size_t bits_bytes = (size_t)e->bits_words * sizeof(unsigned);
  for (int k = 0; k < 2; k++) {
    DSLAM_HIP(hipMalloc(&e->bits_q1[k], bits_bytes));
    DSLAM_HIP(hipMalloc(&e->bits_q2[k], bits_bytes));
    DSLAM_HIP(hipMalloc(&e->bits_mark[k], bits_bytes));
    DSLAM_HIP(hipMemsetAsync(e->bits_q1[k], 0, bits_bytes, e->stream));
    DSLAM_HIP(hipMemsetAsync(e->bits_q2[k], 0, bits_bytes, e->stream));
    DSLAM_HIP(hipMemsetAsync(e->bits_mark[k], 0, bits_bytes, e->stream));
    e->bits_dirty[k] = 0;
  }
  DSLAM_HIP(hipMalloc(&e->bits_retest, bits_bytes));
  DSLAM_HIP(hipMemsetAsync(e->bits_retest, 0, bits_bytes, e->stream));
  int tiles = num_tiles(N > L ? N : L);
  if (tiles < bit_tiles(N) * (kBitTileWords / 32)) tiles = bit_tiles(N) * (kBitTileWords / 32);  // (room for tiles as small as 1024 entries)
  DSLAM_HIP(hipMalloc(&e->agg, (size_t)tiles * 3 * sizeof(unsigned long long)));
  DSLAM_HIP(hipMemsetAsync(e->agg, 0, (size_t)tiles * 3 * sizeof(unsigned long long), e->stream));
  e->agg_tiles = tiles;
  DSLAM_HIP(hipMalloc(&e->tile_counts, (size_t)tiles * 2 * sizeof(int)));
  DSLAM_HIP(hipMalloc(&e->tile_offsets, (size_t)tiles * 2 * sizeof(int)));
  const size_t list_len = (size_t)(N > L ? N : L);
  DSLAM_HIP(hipMalloc(&e->list_a, list_len * sizeof(int)));
  DSLAM_HIP(hipMalloc(&e->list_b, list_len * sizeof(int)));
  DSLAM_HIP(hipMalloc(&e->list_c, list_len * sizeof(int)));
  DSLAM_HIP(hipMalloc(&e->list_d, list_len * sizeof(int)));
  DSLAM_HIP(hipMalloc(&e->pos_scratch, (size_t)L * sizeof(short4)));
  DSLAM_HIP(hipMalloc(&e->rem_flags, (size_t)N));
  DSLAM_HIP(hipMalloc(&e->freed_flags, (size_t)N));
  DSLAM_HIP(hipMalloc(&e->rem_cand, (size_t)L + 16));
  DSLAM_HIP(hipMalloc(&e->maint_flags, 4 * sizeof(int)));
  DSLAM_HIP(hipMemsetAsync(e->rem_flags, 0, (size_t)N, e->stream));
  DSLAM_HIP(hipMemsetAsync(e->freed_flags, 0, (size_t)N, e->stream));
  DSLAM_HIP(hipMemsetAsync(e->rem_cand, 0, (size_t)L + 16, e->stream));
  DSLAM_HIP(hipMemsetAsync(e->maint_flags, 0, 4 * sizeof(int), e->stream));
  DSLAM_HIP(hipStreamSynchronize(e->stream));
  e->scratch_entries = N;
  e->scratch_local_blocks = L;
  return DSLAM_OK;
}

static int ensure_staging(dslam_engine *e, size_t bytes) {
  if (bytes <= e->staging_bytes) return DSLAM_OK;
  DSLAM_HIP(hipStreamSynchronize(e->stream));
  if (e->staging_dev) (void)hipFree(e->staging_dev);
  if (e->staging_host) (void)hipHostFree(e->staging_host);
  DSLAM_HIP(hipMalloc(&e->staging_dev, bytes));
  DSLAM_HIP(hipHostMalloc(&e->staging_host, bytes, hipHostMallocDefault));
  e->staging_bytes = bytes;
  return DSLAM_OK;
}

}  // namespace dslam

using namespace dslam;

extern "C" {

const char *dslam_last_error(void) { return g_last_error.c_str(); }
const char *dslam_version(void) { return "dslam_fusion 0.1 (gfx950)"; }

static int engine_allocate(dslam_engine *e) {
  DSLAM_HIP(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
  DSLAM_HIP(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
  e->pinned_bytes = 64 * 1024;
  DSLAM_HIP(hipHostMalloc(&e->pinned, e->pinned_bytes, hipHostMallocDefault));
  memset(e->pinned, 0, e->pinned_bytes);
  DSLAM_HIP(hipMalloc(&e->misc_counter, 16 * sizeof(int)));
  DSLAM_HIP(hipMalloc(&e->ticket, 16 * sizeof(unsigned)));
  DSLAM_HIP(hipMemset(e->ticket, 0, 16 * sizeof(unsigned)));
  e->ticket_base = 0;
  e->hip_failures_seen = hip_failure_count();
  DSLAM_HIP(hipHostMalloc((void **)&e->err_host, 64, hipHostMallocDefault));
  *e->err_host = 0;
  return DSLAM_OK;
}

int dslam_device_numa_node(int device_index, int *node_out) {
  if (!node_out) return DSLAM_ERR_INVALID;
  *node_out = -1;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    set_last_error("no HIP device: libdslam_fusion has no CPU path");
    return DSLAM_ERR_NO_DEVICE;
  }
  DSLAM_REQUIRE(device_index >= 0 && device_index < n, "device index out of range");
  char bus[64] = {0};
  DSLAM_HIP(hipDeviceGetPCIBusId(bus, (int)sizeof(bus) - 1, device_index));
  for (char *c = bus; *c; c++) *c = (char)tolower(*c);
  char path[160];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
  if (FILE *f = fopen(path, "r")) {
    int node = -1;
    if (fscanf(f, "%d", &node) == 1) *node_out = node;
    fclose(f);
  }
  return DSLAM_OK;   // (-1: the platform does not say)
}

int dslam_engine_create(int device_index, dslam_engine **out) {
  if (!out) return DSLAM_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    set_last_error("no HIP device: libdslam_fusion has no CPU path");
    return DSLAM_ERR_NO_DEVICE;
  }
  DSLAM_REQUIRE(device_index >= 0 && device_index < n, "device index out of range");
  DSLAM_HIP(hipSetDevice(device_index));
  dslam_engine *e = new dslam_engine();
  e->device = device_index;
  const int rc = engine_allocate(e);
  if (rc) {
    (void)dslam_engine_destroy(e);
    return rc;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_index) == hipSuccess) e->sm_count = prop.multiProcessorCount;
  *out = e;
  return DSLAM_OK;
}

int dslam_engine_destroy(dslam_engine *e) {
  if (!e) return DSLAM_OK;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  if (e->copy_stream) { (void)hipStreamSynchronize(e->copy_stream); (void)hipStreamDestroy(e->copy_stream); }
  free_dev(e->order_keys); free_dev(e->alloc_type); free_dev(e->block_coords); free_dev(e->tile_counts); free_dev(e->tile_offsets);
  free_dev(e->list_a); free_dev(e->list_b); free_dev(e->list_c); free_dev(e->list_d); free_dev(e->pos_scratch);
  free_dev(e->agg); free_dev(e->ticket);
  for (int k = 0; k < 2; k++) { free_dev(e->bits_q1[k]); free_dev(e->bits_q2[k]); free_dev(e->bits_mark[k]); }
  free_dev(e->bits_retest);
  free_dev(e->rem_flags); free_dev(e->freed_flags); free_dev(e->rem_cand); free_dev(e->maint_flags);
  if (e->staging_dev) (void)hipFree(e->staging_dev);
  if (e->staging_host) (void)hipHostFree(e->staging_host);
  if (e->pinned) (void)hipHostFree(e->pinned);
  if (e->err_host) (void)hipHostFree(e->err_host);
  if (e->timer_counts_dev) (void)hipFree(e->timer_counts_dev);
  free_dev(e->misc_counter);
  free_dev(e->mesh_positions); free_dev(e->mesh_colours);
  if (e->icp_partials_host) (void)hipHostFree(e->icp_partials_host);  // (icp_partials is its device alias)
  for (auto ev : e->ev_pool) (void)hipEventDestroy(ev);
  for (dslam_fence *f : e->fences) {
    if (f->zombie) { if (f->ev) (void)hipEventDestroy(f->ev); delete f; }
    else f->engine = nullptr;   // the caller still holds it: its destroy call must not look for this engine
  }
  e->fences.clear();
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
  return DSLAM_OK;
}

int dslam_selftest_division(dslam_engine *e, long long samples, long long *mismatches_out) {
  DSLAM_REQUIRE(e && mismatches_out && samples > 0, "bad argument");
  unsigned long long *dev = reinterpret_cast<unsigned long long *>(e->misc_counter + 4);  // 8-byte aligned slot
  int rc = launch_selftest_division(e, samples, dev);
  if (rc) return rc;
  DSLAM_HIP(hipMemcpyAsync(e->pinned, dev, sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
  DSLAM_HIP(hipStreamSynchronize(e->stream));
  *mismatches_out = (long long)*reinterpret_cast<unsigned long long *>(e->pinned);
  return DSLAM_OK;
}

int dslam_debug_set_render_tile_budget(dslam_engine *e, int budget) {
  DSLAM_REQUIRE(e && budget > 0, "bad argument");
  e->render_tile_budget = budget;
  return DSLAM_OK;
}

int dslam_debug_set_push_job_min(dslam_engine *e, int min_visible_blocks) {
  DSLAM_REQUIRE(e && min_visible_blocks >= 0, "bad argument");
  e->push_job_min = min_visible_blocks;
  return DSLAM_OK;
}

int dslam_debug_stream_launches(dslam_engine *e, long long *count_out) {
  DSLAM_REQUIRE(e && count_out, "bad argument");
  *count_out = e->stream_launches;
  return DSLAM_OK;
}

int dslam_debug_inject_device_error(dslam_engine *e, dslam_scene *s, int bits) {
  DSLAM_REQUIRE(e && s && s->engine == e && (bits == 1 || bits == 2 || bits == 3), "bad argument");
  int rc = launch_inject_error(e, s, bits);
  if (rc) return rc;
  return finish_call(e);
}

int dslam_engine_set_async(dslam_engine *e, int async_mode) {
  DSLAM_REQUIRE(e, "null engine");
  e->async_mode = async_mode != 0;
  return DSLAM_OK;
}
int dslam_engine_synchronize(dslam_engine *e) {
  DSLAM_REQUIRE(e, "null engine");
  return sync_check(e);  // (every pipelined upload is waited for by a kernel of this stream)
}

// ---- fences --------------------------------------------------------------------------------------------------
int dslam_fence_create(dslam_engine *e, dslam_fence **out) {
  DSLAM_REQUIRE(e && out, "null argument");
  *out = nullptr;
  dslam_fence *f = new dslam_fence();
  f->engine = e;
  const hipError_t err = hipEventCreateWithFlags(&f->ev, hipEventDisableTiming);
  if (err != hipSuccess) { delete f; return hip_fail(err, "hipEventCreateWithFlags", __FILE__, __LINE__); }
  e->fences.push_back(f);
  *out = f;
  return DSLAM_OK;
}
static void fence_free(dslam_fence *f) {
  if (f->engine) {
    auto &v = f->engine->fences;
    v.erase(std::remove(v.begin(), v.end(), f), v.end());
  }
  if (f->ev) (void)hipEventDestroy(f->ev);
  delete f;
}
// a view stops waiting on the fence event it had borrowed for landing buffer b
static void release_lender(dslam_view *v, int b) {
  dslam_fence *f = v->up_lender[b];
  v->up_lender[b] = nullptr;
  if (f && --f->lent_count == 0 && f->zombie) fence_free(f);
}
int dslam_fence_destroy(dslam_fence *f) {
  if (!f) return DSLAM_OK;
  if (f->engine && f->engine->last_fence == f) f->engine->last_fence = nullptr;
  if (f->engine && f->lent_count > 0) { f->zombie = true; return DSLAM_OK; }  // (a view still waits on its event)
  fence_free(f);
  return DSLAM_OK;
}
int dslam_fence_record(dslam_engine *e, dslam_fence *f) {
  DSLAM_REQUIRE(e && f && f->engine == e, "fence belongs to a different engine");
  DSLAM_HIP(hipEventRecord(f->ev, e->stream));
  f->recorded = true;
  f->view_reads_at_record = e->view_reads;
  e->last_fence = f;
  return DSLAM_OK;
}
int dslam_fence_wait(dslam_fence *f) {
  DSLAM_REQUIRE(f, "null fence");
  if (f->recorded) DSLAM_HIP(hipEventSynchronize(f->ev));
  return f->engine ? device_errors(f->engine) : DSLAM_OK;   // (what kernels in front of the fence reported, see sync_check)
}
int dslam_fence_query(dslam_fence *f, int *done) {
  DSLAM_REQUIRE(f && done, "null argument");
  *done = 1;
  if (f->recorded) {
    const hipError_t err = hipEventQuery(f->ev);
    if (err == hipErrorNotReady) *done = 0;
    else if (err != hipSuccess) return hip_fail(err, "hipEventQuery", __FILE__, __LINE__);
  }
  return DSLAM_OK;
}
void *dslam_engine_stream(dslam_engine *e) { return e ? (void *)e->stream : nullptr; }

// ---- scene ---------------------------------------------------------------------------------------------------
// every device / pinned allocation of a scene; on failure the caller destroys the half-built object (the destroy
// function frees whatever is there), so no error path leaks
static int scene_allocate(dslam_engine *e, dslam_scene *s, void *ext_voxels) {
  const size_t vox_bytes = (size_t)s->p.num_local_blocks * kBlock3 * sizeof(uint2);
  DSLAM_HIP(hipMalloc(&s->hash, (size_t)s->n_entries * sizeof(HashEntry)));
  if (ext_voxels) {
    s->voxels = reinterpret_cast<uint2 *>(ext_voxels);
    s->voxels_external = true;
  } else {
    DSLAM_HIP(hipMalloc(&s->voxels, vox_bytes));
  }
  DSLAM_HIP(hipMalloc(&s->alloc_list, (size_t)s->p.num_local_blocks * sizeof(int)));
  DSLAM_HIP(hipMalloc(&s->excess_list, (size_t)s->p.num_excess * sizeof(int)));
  DSLAM_HIP(hipMalloc(&s->last_seen, (size_t)s->p.num_local_blocks * sizeof(int)));
  DSLAM_HIP(hipMalloc(&s->masks, (size_t)s->p.num_local_blocks * 2 * s->history_words * sizeof(unsigned long long)));
  DSLAM_HIP(hipMalloc(&s->counters, sizeof(SceneCounters)));
  DSLAM_HIP(hipMalloc(&s->alloc_bits, (size_t)bit_tiles(s->n_entries) * kBitTileWords * sizeof(unsigned)));
  if (s->p.use_swapping) {
    DSLAM_HIP(hipMalloc(&s->swap_state, s->n_entries));
    DSLAM_HIP(hipMalloc(&s->swap1_bits, (size_t)bit_tiles(s->n_entries) * kBitTileWords * sizeof(unsigned)));
    DSLAM_HIP(hipMalloc(&s->slot_dev, (size_t)s->n_entries * sizeof(int)));
    DSLAM_HIP(hipMemsetAsync(s->slot_dev, 0xff, (size_t)s->n_entries * sizeof(int), e->stream));  // -1 everywhere
    DSLAM_HIP(hipHostMalloc((void **)&s->next_slot_host, 64, hipHostMallocDefault));
    *s->next_slot_host = 0;
    DSLAM_HIP(hipMalloc(&s->slab_ptrs_dev, (size_t)kMaxSlabs * sizeof(uint4 *)));
  }
  int rc = ensure_scratch(e, s->n_entries, s->p.num_local_blocks);
  if (rc) return rc;
  if ((rc = launch_scene_reset(e, s))) return rc;
  DSLAM_HIP(hipStreamSynchronize(e->stream));
  return DSLAM_OK;
}

int dslam_scene_create(dslam_engine *e, const dslam_scene_params *p, void *ext_voxels, dslam_scene **out) {
  DSLAM_REQUIRE(e && p && out, "null argument");
  *out = nullptr;
  DSLAM_HIP(hipSetDevice(e->device));
  dslam_scene *s = new dslam_scene();
  s->version = next_map_version();
  s->engine = e;
  s->p = *p;
  if (s->p.num_local_blocks <= 0) s->p.num_local_blocks = DSLAM_DEFAULT_LOCAL_BLOCK_NUM;
  if (s->p.num_buckets <= 0) s->p.num_buckets = DSLAM_DEFAULT_BUCKET_NUM;
  if (s->p.num_excess <= 0) s->p.num_excess = DSLAM_DEFAULT_EXCESS_LIST_SIZE;
  if (s->p.history_words <= 0) s->p.history_words = 4;
  s->history_words = s->p.history_words;
  if ((s->p.num_buckets & (s->p.num_buckets - 1)) || ((s->p.num_buckets + s->p.num_excess) & 15) ||
      s->p.max_w < 1 || s->p.max_w > 255 || !(s->p.voxel_size > 0) || !(s->p.mu > 0)) {
    delete s;
    set_last_error("invalid scene parameters (buckets must be a power of two, entries a multiple of 16, 1<=max_w<=255)");
    return DSLAM_ERR_INVALID;
  }
  s->n_entries = s->p.num_buckets + s->p.num_excess;
  const int rc = scene_allocate(e, s, ext_voxels);
  if (rc) {
    (void)dslam_scene_destroy(s);  // frees whatever was allocated before the failure (last_error is kept)
    return rc;
  }
  *out = s;
  return DSLAM_OK;
}

int dslam_scene_destroy(dslam_scene *s) {
  if (!s) return DSLAM_OK;
  (void)hipStreamSynchronize(s->engine->stream);
  free_dev(s->hash);
  if (!s->voxels_external) free_dev(s->voxels);
  free_dev(s->alloc_list); free_dev(s->excess_list); free_dev(s->last_seen); free_dev(s->masks); free_dev(s->counters);
  free_dev(s->swap_state); free_dev(s->slab_ptrs_dev); free_dev(s->alloc_bits); free_dev(s->swap1_bits);
  free_dev(s->dirty); free_dev(s->dirty_list); free_dev(s->dirty_counts);
  free_dev(s->batch_depth);
  free_dev(s->batch_born); free_dev(s->batch_opmask); free_dev(s->batch_slot_entry); free_dev(s->batch_marks); free_dev(s->batch_order); free_dev(s->batch_counters);
  if (s->batch_ops_dev) (void)hipFree(s->batch_ops_dev);
  if (s->batch_lists_dev) (void)hipFree(s->batch_lists_dev);
  if (s->batch_staging) (void)hipHostFree(s->batch_staging);
  if (s->batch_staging_ev) (void)hipEventDestroy(s->batch_staging_ev);
  for (uint4 *slab : s->slabs) (void)hipHostFree(slab);
  if (s->next_slot_host) (void)hipHostFree(s->next_slot_host);
  free_dev(s->slot_dev);
  delete s;
  return DSLAM_OK;
}

int dslam_scene_reset(dslam_engine *e, dslam_scene *s) {
  DSLAM_REQUIRE(e && s, "null argument");
  s->version = next_map_version();  // the map may change: GetImage memos of this scene are stale
  int rc = launch_scene_reset(e, s);
  if (rc) return rc;
  for (int q = 0; q < 2; q++) { s->ring_head[q] = 0; s->ring_next[q] = 0; s->decay_cursor[q] = 0; }
  s->frame_counter = 0;
  if (s->slot_dev) {  // the slabs stay; their slots are dealt again from the start (the counters were just zeroed)
    DSLAM_HIP(hipMemsetAsync(s->slot_dev, 0xff, (size_t)s->n_entries * sizeof(int), e->stream));
    DSLAM_HIP(hipStreamSynchronize(e->stream));
    *s->next_slot_host = 0;
    s->slot_bound = 0;
  }
  return finish_call(e);
}

int dslam_scene_get_params(const dslam_scene *s, dslam_scene_params *out) {
  DSLAM_REQUIRE(s && out, "null argument");
  *out = s->p;
  return DSLAM_OK;
}

int dslam_scene_set_shard(dslam_scene *s, int shard, int num_shards, int chunk_blocks) {
  DSLAM_REQUIRE(s && num_shards >= 1 && shard >= 0 && shard < num_shards && chunk_blocks >= 1, "bad shard spec");
  // A rank that does not own a block still swaps its (stale) copy of it out to ITS host store during the batch, and
  // the exchange moves device blocks only: the ranks' global caches would drift apart.  Refused, not silently wrong.
  DSLAM_REQUIRE(!(s->p.use_swapping && num_shards > 1), "a scene with host swapping cannot be sharded (the host store is per rank)");
  s->shard = shard; s->num_shards = num_shards; s->chunk_blocks = chunk_blocks;
  return DSLAM_OK;
}

int dslam_scene_set_shard_range(dslam_scene *s, int first_block, int num_blocks) {
  DSLAM_REQUIRE(s && first_block >= 0, "bad shard range");
  DSLAM_REQUIRE(!(s->p.use_swapping && num_blocks >= 0), "a scene with host swapping cannot be sharded (the host store is per rank)");
  s->shard_first = first_block; s->shard_count = num_blocks;
  return DSLAM_OK;
}

// ---- which blocks a sharded batch touched, and moving exactly those (csrc/shard.hip) ------------------------------
int dslam_scene_track_dirty(dslam_engine *e, dslam_scene *s, int enable) {
  DSLAM_REQUIRE(e && s, "null argument");
  if (enable) {
    const size_t n = (size_t)s->p.num_local_blocks;
    if (!s->dirty) {
      DSLAM_HIP(hipMalloc(&s->dirty, n));
      DSLAM_HIP(hipMalloc(&s->dirty_list, n * sizeof(int)));
      DSLAM_HIP(hipMalloc(&s->dirty_counts, 128 * sizeof(int)));
    }
    DSLAM_HIP(hipMemsetAsync(s->dirty, 0, n, e->stream));
    s->dirty_shards = 0;
  }
  s->dirty_tracking = enable != 0;
  return finish_call(e);
}

int dslam_shard_dirty_plan(dslam_engine *e, dslam_scene *s, int num_shards, int chunk_blocks, int32_t *counts_out) {
  DSLAM_REQUIRE(e && s && counts_out, "null argument");
  return launch_dirty_plan(e, s, num_shards, chunk_blocks, counts_out);
}

int dslam_shard_dirty_pack(dslam_engine *e, const dslam_scene *s, int shard, void *send_dev, int capacity_blocks) {
  DSLAM_REQUIRE(e && s, "null argument");
  int rc = launch_dirty_pack(e, s, shard, send_dev, capacity_blocks);
  if (rc) return rc;
  return finish_call(e);
}

int dslam_shard_dirty_unpack(dslam_engine *e, dslam_scene *s, int skip_shard, const void *recv_dev, int stride_blocks) {
  DSLAM_REQUIRE(e && s, "null argument");
  s->version = next_map_version();  // the map changes: GetImage memos of this scene are stale
  int rc = launch_dirty_unpack(e, s, skip_shard, recv_dev, stride_blocks);
  if (rc) return rc;
  return finish_call(e);
}

// ---- render state / view --------------------------------------------------------------------------------------
static int render_state_allocate(dslam_engine *e, dslam_render_state *r) {
  const size_t npix = (size_t)r->w * r->h;
  DSLAM_HIP(hipMalloc(&r->visible_ids, (size_t)r->n_local * sizeof(int)));
  DSLAM_HIP(hipHostMalloc((void **)&r->vis_hint, 64, hipHostMallocDefault));
  *r->vis_hint = 0;
  DSLAM_HIP(hipMalloc(&r->visible_type, r->n_entries));
  const size_t vis_bits_bytes = (size_t)bit_tiles(r->n_entries) * kBitTileWords * sizeof(unsigned);
  DSLAM_HIP(hipMalloc(&r->vis_bits, vis_bits_bytes));
  DSLAM_HIP(hipMemsetAsync(r->vis_bits, 0, vis_bits_bytes, e->stream));
  DSLAM_HIP(hipMalloc(&r->range, npix * sizeof(float2)));
  DSLAM_HIP(hipMalloc(&r->raycast, npix * sizeof(float4)));
  DSLAM_HIP(hipMalloc(&r->image_rgba, npix * sizeof(uchar4)));
  DSLAM_HIP(hipMalloc(&r->image_float, npix * sizeof(float)));
  DSLAM_HIP(hipMalloc(&r->proj_boxes, (size_t)r->n_local * sizeof(int4)));
  DSLAM_HIP(hipMalloc(&r->proj_z, (size_t)r->n_local * sizeof(float2)));
  DSLAM_HIP(hipMalloc(&r->proj_req, (size_t)r->n_local * sizeof(int)));
  DSLAM_HIP(hipMalloc(&r->proj_wg_tiles, (size_t)(r->n_entries / 1024 + 1024) * sizeof(int)));
  DSLAM_HIP(hipMalloc(&r->counters, sizeof(RenderCounters)));
  DSLAM_HIP(hipMemsetAsync(r->visible_type, 0, r->n_entries, e->stream));
  DSLAM_HIP(hipMemsetAsync(r->counters, 0, sizeof(RenderCounters), e->stream));
  DSLAM_HIP(hipMemsetAsync(r->range, 0, npix * sizeof(float2), e->stream));
  DSLAM_HIP(hipMemsetAsync(r->raycast, 0, npix * sizeof(float4), e->stream));
  DSLAM_HIP(hipMemsetAsync(r->image_rgba, 0, npix * sizeof(uchar4), e->stream));
  DSLAM_HIP(hipMemsetAsync(r->image_float, 0, npix * sizeof(float), e->stream));
  DSLAM_HIP(hipStreamSynchronize(e->stream));
  return DSLAM_OK;
}

int dslam_render_state_create(dslam_engine *e, const dslam_scene *s, int w, int h, dslam_render_state **out) {
  DSLAM_REQUIRE(e && s && out && w > 0 && h > 0, "bad argument");
  *out = nullptr;
  dslam_render_state *r = new dslam_render_state();
  r->engine = e; r->w = w; r->h = h; r->n_entries = s->n_entries; r->n_local = s->p.num_local_blocks;
  const int rc = render_state_allocate(e, r);
  if (rc) {
    (void)dslam_render_state_destroy(r);
    return rc;
  }
  *out = r;
  return DSLAM_OK;
}

int dslam_render_state_destroy(dslam_render_state *r) {
  if (!r) return DSLAM_OK;
  (void)hipStreamSynchronize(r->engine->stream);
  free_dev(r->visible_ids); free_dev(r->visible_type); free_dev(r->vis_bits); free_dev(r->range); free_dev(r->raycast);
  free_dev(r->image_rgba); free_dev(r->image_float); free_dev(r->icp_points); free_dev(r->icp_normals);
  free_dev(r->raycast_image);
  free_dev(r->proj_boxes); free_dev(r->proj_z); free_dev(r->proj_req); free_dev(r->proj_wg_tiles); free_dev(r->counters);
  if (r->vis_hint) (void)hipHostFree(r->vis_hint);
  delete r;
  return DSLAM_OK;
}

static int view_allocate(dslam_engine *e, dslam_view *v) {
  const int w_rgb = v->w_rgb, h_rgb = v->h_rgb, w_d = v->w_d, h_d = v->h_d;
  // (the RGBA image and the int16 depth image in ONE allocation, the depth image at the next 256-byte boundary: a frame whose two
  // host images sit back to back the same way -- every 640x480 frame out of dslam_host_alloc -- goes up as one copy)
  const size_t c_round = ((size_t)w_rgb * h_rgb * sizeof(uchar4) + 255) & ~(size_t)255;
  DSLAM_HIP(hipMalloc(&v->rgba, c_round + (size_t)w_d * h_d * sizeof(short)));
  v->raw_depth = reinterpret_cast<short *>(reinterpret_cast<char *>(v->rgba) + c_round);
  DSLAM_HIP(hipMalloc(&v->depth, (size_t)w_d * h_d * sizeof(float)));
  v->rgba_src = v->rgba; v->raw_src = v->raw_depth;
  DSLAM_HIP(hipMemsetAsync(v->rgba, 0, (size_t)w_rgb * h_rgb * sizeof(uchar4), e->stream));
  DSLAM_HIP(hipMemsetAsync(v->depth, 0, (size_t)w_d * h_d * sizeof(float), e->stream));
  DSLAM_HIP(hipMemsetAsync(v->raw_depth, 0, (size_t)w_d * h_d * sizeof(short), e->stream));
  DSLAM_HIP(hipStreamSynchronize(e->stream));
  return DSLAM_OK;
}

int dslam_view_create(dslam_engine *e, int w_rgb, int h_rgb, int w_d, int h_d, dslam_view **out) {
  DSLAM_REQUIRE(e && out && w_rgb > 0 && h_rgb > 0 && w_d > 0 && h_d > 0, "bad argument");
  *out = nullptr;
  dslam_view *v = new dslam_view();
  v->engine = e; v->w_rgb = w_rgb; v->h_rgb = h_rgb; v->w_d = w_d; v->h_d = h_d;
  const int rc = view_allocate(e, v);
  if (rc) {
    (void)dslam_view_destroy(v);
    return rc;
  }
  *out = v;
  return DSLAM_OK;
}

int dslam_view_destroy(dslam_view *v) {
  if (!v) return DSLAM_OK;
  (void)hipStreamSynchronize(v->engine->stream);
  if (v->engine->copy_stream) (void)hipStreamSynchronize(v->engine->copy_stream);
  free_dev(v->rgba); free_dev(v->depth); free_dev(v->pyramid);   // (raw_depth lives in rgba's allocation)
  for (int b = 0; b < 2; b++) {
    release_lender(v, b);
    free_dev(v->up_rgba[b]);  // (up_raw[b] points into the same allocation)
    if (v->up_done[b]) (void)hipEventDestroy(v->up_done[b]);
    if (v->up_consumed[b]) (void)hipEventDestroy(v->up_consumed[b]);
  }
  delete v;
  return DSLAM_OK;
}

// common tail of the UpdateView entry points: remember the sources; with useBilateralFilter the float depth image
// is produced now (conversion + five filter passes), otherwise lazily by the next consumer
static int finish_view_update(dslam_engine *e, dslam_view *v, const void *rgba_dev, const void *depth_dev, float a,
                              float b, double timestamp, int use_bilateral) {
  int rc = launch_view_convert(e, v, rgba_dev, depth_dev, a, b);
  if (rc) return rc;
  if (use_bilateral) {
    DSLAM_REQUIRE(v->w_d >= 5 && v->h_d >= 5, "bilateral filter needs an image of at least 5x5");
    if ((rc = launch_bilateral(e, v))) return rc;
    e->view_reads++;  // (the filter reads the landing buffer: a fence recorded before it does not cover it)
  }
  v->timestamp = timestamp;
  return finish_call(e);
}

// ---- page-locked host buffers for the caller's input images ---------------------------------------------------
// Upstream ORUtils::MemoryBlock allocates its host side with cudaMallocHost whenever the block also has a device
// side; dslam_host_alloc gives the ITMLib mirror the same thing without it having to link the HIP runtime.  The
// engine remembers the ranges, so an upload from one of them can skip the staging copy.
static std::mutex g_pinned_mu;
static std::vector<std::pair<const char *, size_t>> g_pinned_ranges;
// Buffers of up to kArenaMax bytes are carved out of page-locked arenas one behind the other (256-byte granules), so that
// images a caller allocates in a row -- InfiniTamDriver's rgb_itm_ and raw_depth_itm_ (InfiniTamDriver.h:103-104), the bench's
// frame ring -- are CONTIGUOUS in host memory: dslam_view_update then moves a frame with one copy instead of two (one DMA
// start-up less per frame: what the pipelined upload path already did for callers that laid their frames out that way by
// hand).  An arena goes back to the runtime when its last buffer is freed.
namespace {
constexpr size_t kArenaMax = 8u << 20, kArenaBytes = 32u << 20;
struct PinnedArena { char *base; size_t used; int live; };
std::vector<PinnedArena> g_arenas;
}  // namespace

int dslam_host_alloc(size_t bytes, void **out) {
  DSLAM_REQUIRE(out, "null argument");
  const size_t need = ((bytes ? bytes : 1) + 255) & ~(size_t)255;
  std::lock_guard<std::mutex> lock(g_pinned_mu);
  void *p = nullptr;
  if (need <= kArenaMax) {
    if (g_arenas.empty() || g_arenas.back().used + need > kArenaBytes) {
      void *base = nullptr;
      DSLAM_HIP(hipHostMalloc(&base, kArenaBytes, hipHostMallocDefault));
      g_arenas.push_back({static_cast<char *>(base), 0, 0});
    }
    PinnedArena &a = g_arenas.back();
    p = a.base + a.used;
    a.used += need;
    a.live++;
  } else {
    DSLAM_HIP(hipHostMalloc(&p, need, hipHostMallocDefault));
  }
  memset(p, 0, bytes);
  g_pinned_ranges.emplace_back(static_cast<const char *>(p), bytes ? bytes : 1);
  *out = p;
  return DSLAM_OK;
}

int dslam_host_free(void *p) {
  if (!p) return DSLAM_OK;
  std::lock_guard<std::mutex> lock(g_pinned_mu);
  size_t i = 0;
  while (i < g_pinned_ranges.size() && g_pinned_ranges[i].first != p) i++;
  DSLAM_REQUIRE(i < g_pinned_ranges.size(), "dslam_host_free: pointer was not returned by dslam_host_alloc");
  g_pinned_ranges.erase(g_pinned_ranges.begin() + i);
  for (size_t k = 0; k < g_arenas.size(); k++) {
    PinnedArena &a = g_arenas[k];
    if (static_cast<char *>(p) >= a.base && static_cast<char *>(p) < a.base + kArenaBytes) {
      if (--a.live == 0) {   // (the arena that is being filled is kept while it has room: its next buffer may come at once)
        if (k + 1 == g_arenas.size() && a.used < kArenaBytes) { a.used = 0; return DSLAM_OK; }
        char *base = a.base;
        g_arenas.erase(g_arenas.begin() + k);
        DSLAM_HIP(hipHostFree(base));
      }
      return DSLAM_OK;
    }
  }
  DSLAM_HIP(hipHostFree(p));
  return DSLAM_OK;
}

static bool in_pinned_range(const void *p, size_t bytes) {
  const char *c = static_cast<const char *>(p);
  std::lock_guard<std::mutex> lock(g_pinned_mu);
  for (const auto &r : g_pinned_ranges)
    if (c >= r.first && c + bytes <= r.first + r.second) return true;
  return false;
}

// Async engine + page-locked RGBA / depth images: the copy runs on the engine's copy stream into landing buffer b of
// the view while the compute stream is still reading buffer b ^ 1 (the previous frame's kernels, enqueued earlier).
//   up_consumed[b]  recorded on the compute stream at the NEXT update, i.e. behind every kernel that reads buffer b;
//                   the copy that refills b (two updates later) may only start once it has passed
//   up_done[b]      recorded on the copy stream behind the copy; this frame's kernels may only start once it has passed
// Both conditions are awaited by the HOST (the calling thread sits out the copy, ~40 us per 640x480 frame, while the
// GPU works on the previous frame), not by hipStreamWaitEvent: on this runtime a cross-stream wait in front of a
// frame's kernels cost the compute stream ~30 us per frame (197 vs 167 us per step), more than the copy it hides;
// DSLAM_PIPELINE_STREAM_WAITS=1 selects that variant for measurement.
// *rgba_out / *raw_out receive the buffers the view reads from now on.
static int upload_view_pipelined(dslam_engine *e, dslam_view *v, const uint8_t *rgba_host, const int16_t *depth_host,
                                 const void **rgba_out, const void **raw_out) {
  static const bool stream_waits = getenv("DSLAM_PIPELINE_STREAM_WAITS") && atoi(getenv("DSLAM_PIPELINE_STREAM_WAITS")) != 0;
  const size_t c_bytes = (size_t)v->w_rgb * v->h_rgb * 4, d_bytes = (size_t)v->w_d * v->h_d * 2;
  if (!v->up_rgba[0]) {
    for (int b = 0; b < 2; b++) {
      // one allocation per landing buffer (RGBA image, then the depth image): a caller that keeps a frame's two images
      // back to back gets ONE copy per frame
      DSLAM_HIP(hipMalloc(&v->up_rgba[b], c_bytes + d_bytes));
      v->up_raw[b] = reinterpret_cast<short *>(reinterpret_cast<char *>(v->up_rgba[b]) + c_bytes);
      DSLAM_HIP(hipEventCreateWithFlags(&v->up_done[b], hipEventDisableTiming));
      DSLAM_HIP(hipEventCreateWithFlags(&v->up_consumed[b], hipEventDisableTiming));
    }
  }
  const int b = v->up_next;
  v->up_next ^= 1;
  if (v->up_used[b ^ 1]) {
    dslam_fence *f = e->last_fence;
    release_lender(v, b ^ 1);
    if (f && f->recorded && f->view_reads_at_record == e->view_reads) {
      // the caller's fence sits behind every kernel that read buffer b ^ 1 (no view was read since it was recorded)
      f->lent_count++;
      v->up_lender[b ^ 1] = f;
      v->up_consumed_by[b ^ 1] = f->ev;
    } else {
      DSLAM_HIP(hipEventRecord(v->up_consumed[b ^ 1], e->stream));
      v->up_consumed_by[b ^ 1] = v->up_consumed[b ^ 1];
    }
  }
  if (v->up_used[b]) {
    if (stream_waits) DSLAM_HIP(hipStreamWaitEvent(e->copy_stream, v->up_consumed_by[b], 0));
    else DSLAM_HIP(hipEventSynchronize(v->up_consumed_by[b]));
  }
  if (reinterpret_cast<const uint8_t *>(depth_host) == rgba_host + c_bytes) {
    DSLAM_HIP(hipMemcpyAsync(v->up_rgba[b], rgba_host, c_bytes + d_bytes, hipMemcpyHostToDevice, e->copy_stream));
  } else {
    DSLAM_HIP(hipMemcpyAsync(v->up_rgba[b], rgba_host, c_bytes, hipMemcpyHostToDevice, e->copy_stream));
    DSLAM_HIP(hipMemcpyAsync(v->up_raw[b], depth_host, d_bytes, hipMemcpyHostToDevice, e->copy_stream));
  }
  if (stream_waits) {
    DSLAM_HIP(hipEventRecord(v->up_done[b], e->copy_stream));
    DSLAM_HIP(hipStreamWaitEvent(e->stream, v->up_done[b], 0));
  } else {
    // (the copy stream holds nothing but this frame's copy; waiting for the STREAM returns 8 us earlier than recording an event
    // behind the copy and waiting for that: 41 against 49 us for the 1.84 MB of a 640x480 frame, profiles/experiments/upload_bench.hip)
    DSLAM_HIP(hipStreamSynchronize(e->copy_stream));
  }
  v->up_used[b] = true;
  *rgba_out = v->up_rgba[b];
  *raw_out = v->up_raw[b];
  return DSLAM_OK;
}

static int upload_view_host(dslam_engine *e, dslam_view *v, const uint8_t *colour_host, int colour_channels,
                            const int16_t *depth_host) {
  const size_t c_bytes = (size_t)v->w_rgb * v->h_rgb * colour_channels, d_bytes = (size_t)v->w_d * v->h_d * 2;
  int rc = ensure_staging(e, (size_t)v->w_rgb * v->h_rgb * 4 + d_bytes);
  if (rc) return rc;
  const void *colour_src, *depth_src;
  if (!e->async_mode && in_pinned_range(colour_host, c_bytes) && in_pinned_range(depth_host, d_bytes)) {
    // page-locked caller buffers and a call that only returns once the stream has drained: DMA straight from them
    colour_src = colour_host; depth_src = depth_host;
  } else {
    // the caller may reuse its buffers right after the call (CvToItm rewrites them every frame), so stage through
    // pinned memory; in async mode the previous upload must have drained before the staging buffer is rewritten
    DSLAM_HIP(hipStreamSynchronize(e->stream));
    memcpy(e->staging_host, colour_host, c_bytes);
    memcpy((char *)e->staging_host + c_bytes, depth_host, d_bytes);
    colour_src = e->staging_host; depth_src = (char *)e->staging_host + c_bytes;
  }
  void *colour_dst = colour_channels == 4 ? (void *)v->rgba : e->staging_dev;
  if (colour_channels == 4 && static_cast<const char *>(depth_src) == static_cast<const char *>(colour_src) + c_bytes &&
      reinterpret_cast<char *>(v->raw_depth) == reinterpret_cast<char *>(v->rgba) + c_bytes) {
    DSLAM_HIP(hipMemcpyAsync(v->rgba, colour_src, c_bytes + d_bytes, hipMemcpyHostToDevice, e->stream));   // (one DMA instead of two: -8 us)
  } else {
    DSLAM_HIP(hipMemcpyAsync(colour_dst, colour_src, c_bytes, hipMemcpyHostToDevice, e->stream));
    DSLAM_HIP(hipMemcpyAsync(v->raw_depth, depth_src, d_bytes, hipMemcpyHostToDevice, e->stream));
  }
  if (colour_channels == 3) return launch_bgr_to_rgba(e, e->staging_dev, v->rgba, v->w_rgb * v->h_rgb);
  return DSLAM_OK;
}

int dslam_view_update(dslam_engine *e, dslam_view *v, const uint8_t *rgba_host, const int16_t *depth_host, float a,
                      float b, double timestamp, int use_bilateral) {
  DSLAM_REQUIRE(e && v && rgba_host && depth_host, "null argument");
  if (e->async_mode && in_pinned_range(rgba_host, (size_t)v->w_rgb * v->h_rgb * 4) &&
      in_pinned_range(depth_host, (size_t)v->w_d * v->h_d * 2)) {
    const void *rgba_dev = nullptr, *raw_dev = nullptr;
    int rc = upload_view_pipelined(e, v, rgba_host, depth_host, &rgba_dev, &raw_dev);
    if (rc) return rc;
    return finish_view_update(e, v, rgba_dev, raw_dev, a, b, timestamp, use_bilateral);
  }
  int rc = upload_view_host(e, v, rgba_host, 4, depth_host);
  if (rc) return rc;
  return finish_view_update(e, v, v->rgba, v->raw_depth, a, b, timestamp, use_bilateral);
}

int dslam_view_update_device(dslam_engine *e, dslam_view *v, const void *rgba_dev, const void *depth_dev, float a,
                             float b, double timestamp, int use_bilateral) {
  DSLAM_REQUIRE(e && v && rgba_dev && depth_dev, "null argument");
  return finish_view_update(e, v, rgba_dev, depth_dev, a, b, timestamp, use_bilateral);
}

int dslam_view_update_bgr(dslam_engine *e, dslam_view *v, const uint8_t *bgr_host, const int16_t *depth_host, float a,
                          float b, double timestamp, int use_bilateral) {
  DSLAM_REQUIRE(e && v && bgr_host && depth_host, "null argument");
  int rc = upload_view_host(e, v, bgr_host, 3, depth_host);
  if (rc) return rc;
  return finish_view_update(e, v, v->rgba, v->raw_depth, a, b, timestamp, use_bilateral);
}

int dslam_view_update_bgr_device(dslam_engine *e, dslam_view *v, const void *bgr_dev, const void *depth_dev, float a,
                                 float b, double timestamp, int use_bilateral) {
  DSLAM_REQUIRE(e && v && bgr_dev && depth_dev, "null argument");
  DSLAM_REQUIRE(((uintptr_t)bgr_dev & 3) == 0, "bgr image must be 4-byte aligned");
  int rc = launch_bgr_to_rgba(e, bgr_dev, v->rgba, v->w_rgb * v->h_rgb);
  if (rc) return rc;
  return finish_view_update(e, v, v->rgba, depth_dev, a, b, timestamp, use_bilateral);
}

int dslam_view_update_dataset(dslam_engine *e, dslam_view *v, const uint8_t *colour_host, int channels,
                              const int16_t *depth_raw_host, int depth_format, float max_depth_m, float a, float b,
                              double timestamp, int use_bilateral) {
  DSLAM_REQUIRE(e && v && colour_host && depth_raw_host, "null argument");
  DSLAM_REQUIRE(channels == 3 || channels == 4, "colour_channels must be 3 (BGR) or 4 (RGBA)");
  DSLAM_REQUIRE(depth_format >= DSLAM_DEPTH_MM && depth_format <= DSLAM_DEPTH_RGBD_X5, "unknown depth format");
  int rc = upload_view_host(e, v, colour_host, channels, depth_raw_host);
  if (rc) return rc;
  if (depth_format != DSLAM_DEPTH_MM && (rc = launch_dataset_depth(e, v->raw_depth, v->w_d * v->h_d, depth_format, max_depth_m))) return rc;
  return finish_view_update(e, v, v->rgba, v->raw_depth, a, b, timestamp, use_bilateral);
}

int dslam_download_view_raw_depth(dslam_engine *e, const dslam_view *v, int16_t *out) {
  DSLAM_REQUIRE(e && v && out, "null argument");
  e->view_reads++;  // (see dslam_engine::last_fence)
  DSLAM_HIP(hipMemcpyAsync(out, v->raw_src, (size_t)v->w_d * v->h_d * 2, hipMemcpyDeviceToHost, e->stream));
  DSLAM_HIP(hipStreamSynchronize(e->stream));
  return DSLAM_OK;
}

int dslam_download_view_rgba(dslam_engine *e, const dslam_view *v, uint8_t *out) {
  DSLAM_REQUIRE(e && v && out, "null argument");
  e->view_reads++;  // (see dslam_engine::last_fence)
  DSLAM_HIP(hipMemcpyAsync(out, v->rgba_src, (size_t)v->w_rgb * v->h_rgb * 4, hipMemcpyDeviceToHost, e->stream));
  DSLAM_HIP(hipStreamSynchronize(e->stream));
  return DSLAM_OK;
}

// ---- keyframe store (fusion-frame database payload resident in HBM) -------------------------------------------
int dslam_frame_store_create(dslam_engine *e, int w_rgb, int h_rgb, int w_d, int h_d, int capacity, dslam_frame_store **out) {
  DSLAM_REQUIRE(e && out && w_rgb > 0 && h_rgb > 0 && w_d > 0 && h_d > 0 && capacity > 0, "bad argument");
  dslam_frame_store *fs = new dslam_frame_store();
  fs->engine = e; fs->w_rgb = w_rgb; fs->h_rgb = h_rgb; fs->w_d = w_d; fs->h_d = h_d; fs->capacity = capacity;
  // slots are padded to 256 bytes so every image starts on an aligned address whatever the image size
  fs->rgba_bytes = ((size_t)w_rgb * h_rgb * 4 + 255) & ~(size_t)255;
  fs->depth_bytes = ((size_t)w_d * h_d * 2 + 255) & ~(size_t)255;
  hipError_t err = hipMalloc(&fs->rgba, fs->rgba_bytes * capacity);
  if (err == hipSuccess) err = hipMalloc(&fs->depth, fs->depth_bytes * capacity);
  if (err != hipSuccess) {
    free_dev(fs->rgba); free_dev(fs->depth);
    delete fs;
    set_last_error("frame store: out of device memory");
    return DSLAM_ERR_HIP;
  }
  *out = fs;
  return DSLAM_OK;
}

int dslam_frame_store_destroy(dslam_frame_store *fs) {
  if (!fs) return DSLAM_OK;
  (void)hipStreamSynchronize(fs->engine->stream);
  free_dev(fs->rgba); free_dev(fs->depth); free_dev(fs->lists); free_dev(fs->batch_lists);
  delete fs;
  return DSLAM_OK;
}

static int store_slot_ok(const dslam_engine *e, const dslam_frame_store *fs, int slot) {
  DSLAM_REQUIRE(e && fs && fs->engine == e, "frame store belongs to a different engine");
  DSLAM_REQUIRE(slot >= 0 && slot < fs->capacity, "frame store slot out of range");
  return DSLAM_OK;
}

static int store_put_host(dslam_engine *e, dslam_frame_store *fs, int slot, const uint8_t *colour, int channels, const int16_t *depth) {
  int rc = store_slot_ok(e, fs, slot);
  if (rc) return rc;
  DSLAM_REQUIRE(colour && depth, "null argument");
  const size_t c_bytes = (size_t)fs->w_rgb * fs->h_rgb * channels, d_bytes = (size_t)fs->w_d * fs->h_d * 2;
  if ((rc = ensure_staging(e, (size_t)fs->w_rgb * fs->h_rgb * 4 + d_bytes))) return rc;
  DSLAM_HIP(hipStreamSynchronize(e->stream));
  memcpy(e->staging_host, colour, c_bytes);
  memcpy((char *)e->staging_host + c_bytes, depth, d_bytes);
  unsigned char *rgba_dst = fs->rgba + fs->rgba_bytes * slot;
  DSLAM_HIP(hipMemcpyAsync(channels == 4 ? (void *)rgba_dst : e->staging_dev, e->staging_host, c_bytes, hipMemcpyHostToDevice, e->stream));
  DSLAM_HIP(hipMemcpyAsync(fs->depth + fs->depth_bytes * slot, (char *)e->staging_host + c_bytes, d_bytes, hipMemcpyHostToDevice, e->stream));
  if (channels == 3 && (rc = launch_bgr_to_rgba(e, e->staging_dev, (uchar4 *)rgba_dst, fs->w_rgb * fs->h_rgb))) return rc;
  return finish_call(e);
}

int dslam_frame_store_put(dslam_engine *e, dslam_frame_store *fs, int slot, const uint8_t *rgba_host, const int16_t *depth_host) {
  return store_put_host(e, fs, slot, rgba_host, 4, depth_host);
}
int dslam_frame_store_put_bgr(dslam_engine *e, dslam_frame_store *fs, int slot, const uint8_t *bgr_host, const int16_t *depth_host) {
  return store_put_host(e, fs, slot, bgr_host, 3, depth_host);
}

int dslam_frame_store_put_view(dslam_engine *e, dslam_frame_store *fs, int slot, const dslam_view *v) {
  int rc = store_slot_ok(e, fs, slot);
  if (rc) return rc;
  DSLAM_REQUIRE(v && v->engine == e, "null argument");
  e->view_reads++;  // (see dslam_engine::last_fence)
  DSLAM_REQUIRE(v->w_rgb == fs->w_rgb && v->h_rgb == fs->h_rgb && v->w_d == fs->w_d && v->h_d == fs->h_d, "view and frame store sizes differ");
  DSLAM_HIP(hipMemcpyAsync(fs->rgba + fs->rgba_bytes * slot, v->rgba_src, (size_t)fs->w_rgb * fs->h_rgb * 4, hipMemcpyDeviceToDevice, e->stream));
  DSLAM_HIP(hipMemcpyAsync(fs->depth + fs->depth_bytes * slot, v->raw_src, (size_t)fs->w_d * fs->h_d * 2, hipMemcpyDeviceToDevice, e->stream));
  return finish_call(e);
}

int dslam_frame_store_get(dslam_engine *e, const dslam_frame_store *fs, int slot, uint8_t *rgba_out, int16_t *depth_out) {
  int rc = store_slot_ok(e, fs, slot);
  if (rc) return rc;
  if (rgba_out) DSLAM_HIP(hipMemcpyAsync(rgba_out, fs->rgba + fs->rgba_bytes * slot, (size_t)fs->w_rgb * fs->h_rgb * 4, hipMemcpyDeviceToHost, e->stream));
  if (depth_out) DSLAM_HIP(hipMemcpyAsync(depth_out, fs->depth + fs->depth_bytes * slot, (size_t)fs->w_d * fs->h_d * 2, hipMemcpyDeviceToHost, e->stream));
  DSLAM_HIP(hipStreamSynchronize(e->stream));
  return DSLAM_OK;
}

int dslam_frame_store_device_ptrs(const dslam_frame_store *fs, int slot, void **rgba_dev, void **depth_dev) {
  DSLAM_REQUIRE(fs && slot >= 0 && slot < fs->capacity, "frame store slot out of range");
  if (rgba_dev) *rgba_dev = fs->rgba + fs->rgba_bytes * slot;
  if (depth_dev) *depth_dev = fs->depth + fs->depth_bytes * slot;
  return DSLAM_OK;
}

int dslam_view_update_from_store(dslam_engine *e, dslam_view *v, const dslam_frame_store *fs, int slot, float a, float b,
                                 double timestamp, int use_bilateral) {
  int rc = store_slot_ok(e, fs, slot);
  if (rc) return rc;
  DSLAM_REQUIRE(v && v->engine == e, "null argument");
  DSLAM_REQUIRE(v->w_rgb == fs->w_rgb && v->h_rgb == fs->h_rgb && v->w_d == fs->w_d && v->h_d == fs->h_d, "view and frame store sizes differ");
  return finish_view_update(e, v, fs->rgba + fs->rgba_bytes * slot, fs->depth + fs->depth_bytes * slot, a, b, timestamp, use_bilateral);
}

// ---- the visible list of a keyframe's fusion, kept with the keyframe ------------------------------------------------
static const size_t kListHeader = 64;  // (a RenderCounters-shaped count block, padded)

int dslam_frame_store_enable_lists(dslam_engine *e, dslam_frame_store *fs, const dslam_scene *s) {
  DSLAM_REQUIRE(e && fs && s && fs->engine == e, "bad argument");
  if (fs->lists && fs->list_cap >= s->p.num_local_blocks && fs->list_entries == s->n_entries) return DSLAM_OK;
  DSLAM_HIP(hipStreamSynchronize(e->stream));
  free_dev(fs->lists);
  fs->list_cap = s->p.num_local_blocks;
  fs->list_entries = s->n_entries;   // the lists hold entry ids of a table of this size (checked wherever a list is read)
  fs->list_bytes = (kListHeader + (size_t)fs->list_cap * (sizeof(int) + sizeof(short4)) + 255) & ~(size_t)255;
  DSLAM_HIP(hipMalloc(&fs->lists, fs->list_bytes * fs->capacity));
  fs->has_list.assign(fs->capacity, 0);
  fs->list_ptr.resize(fs->capacity);
  for (int i = 0; i < fs->capacity; i++) fs->list_ptr[i] = fs->lists + fs->list_bytes * i;
  free_dev(fs->batch_lists);   // (sized for the old lists)
  fs->batch_list_ptr.clear();
  return DSLAM_OK;
}

static unsigned char *list_slot(const dslam_frame_store *fs, int slot) { return fs->list_ptr[slot]; }

int dslam_frame_store_put_visible_list(dslam_engine *e, dslam_frame_store *fs, int slot, const dslam_scene *s,
                                       const dslam_render_state *r) {
  int rc = store_slot_ok(e, fs, slot);
  if (rc) return rc;
  DSLAM_REQUIRE(s && r && fs->lists, "dslam_frame_store_enable_lists has not been called");
  DSLAM_REQUIRE(fs->list_cap >= r->n_local, "the store's lists are smaller than this render state's visible list");
  DSLAM_REQUIRE(fs->list_entries == s->n_entries && r->n_entries == s->n_entries, "the store's lists were enabled for a table of another size");
  unsigned char *base = list_slot(fs, slot);
  rc = launch_store_visible_list(e, s, r, base, reinterpret_cast<int *>(base + kListHeader),
                                 reinterpret_cast<short4 *>(base + kListHeader + (size_t)fs->list_cap * sizeof(int)), fs->list_cap);
  if (rc) return rc;
  fs->has_list[slot] = 1;
  return finish_call(e);
}

int dslam_deprocess_frame_stored(dslam_engine *e, dslam_scene *s, const dslam_view *v, const dslam_frame_store *fs, int slot,
                                 const float M_d[16], const float intr_d[4], const float M_rgb[16], const float intr_rgb[4]) {
  int rc = store_slot_ok(e, fs, slot);
  if (rc) return rc;
  DSLAM_REQUIRE(s && v && M_d && intr_d && s->engine == e && v->engine == e, "bad argument");
  e->view_reads++;  // (see dslam_engine::last_fence)
  DSLAM_REQUIRE(fs->lists && fs->has_list[slot], "no visible list was stored for this keyframe slot");
  DSLAM_REQUIRE(fs->list_entries == s->n_entries, "the stored list belongs to a table of another size");   // (its ids index this scene's table)
  s->version = next_map_version();  // the map changes: GetImage memos of this scene are stale
  const unsigned char *base = list_slot(fs, slot);
  rc = launch_integrate_list(e, s, v, base, reinterpret_cast<const int *>(base + kListHeader),
                             reinterpret_cast<const short4 *>(base + kListHeader + (size_t)fs->list_cap * sizeof(int)), M_d, intr_d,
                             M_rgb, intr_rgb, true);
  if (rc) return rc;
  return finish_call(e);
}

// ---- the re-integration batch, block-major (integrate.hip) ---------------------------------------------------------------
namespace {
constexpr int kBatchMax = 32;   // keyframes per launch of the block kernel: two operation bits each in a 64-bit mask
struct HostBatchOp {            // = BatchOp (integrate.hip)
  float M[16];
  const void *depth, *rgba;
  int push_bit, push_frame, pad[2];
};
struct HostBatchList {          // = BatchListRef
  const void *count, *ids, *pos;
};
}  // namespace

// A batch that stops after its first mutation (only a HIP failure can do that: every argument was checked before) leaves
// allocation passes applied whose blocks were never de- or re-integrated: the map is neither the old nor the new one.
// The header says so; what can be kept consistent is: the render state's list is re-derived by its next pass, and the
// keyframes of the chunk lose their stored lists (a later batch refuses them instead of de-integrating from stale lists).
static int batch_failed(dslam_scene *s, dslam_render_state *r, dslam_frame_store *fs, const int32_t *slots, int K, int rc) {
  s->alloc_born = nullptr;
  r->types_follow_list = false;
  r->memo_valid = false;
  for (int k = 0; k < K; k++) fs->has_list[slots[k]] = 0;
  return rc;
}

static int batch_scratch(dslam_engine *e, dslam_scene *s, dslam_frame_store *fs) {
  const size_t L = (size_t)s->p.num_local_blocks;
  const size_t npix = (size_t)fs->w_d * fs->h_d;
  if (s->batch_depth_pixels < npix) {   // (1.2 MB per 640x480 keyframe: 39 MB for the 32 of a chunk)
    DSLAM_HIP(hipStreamSynchronize(e->stream));
    free_dev(s->batch_depth);
    s->batch_depth_pixels = 0;
    DSLAM_HIP(hipMalloc(&s->batch_depth, (size_t)kBatchMax * npix * sizeof(float)));
    s->batch_depth_pixels = npix;
  }
  if (!s->batch_born) {
    DSLAM_HIP(hipMalloc(&s->batch_born, L * sizeof(int)));
    DSLAM_HIP(hipMalloc(&s->batch_opmask, L * sizeof(unsigned long long)));
    DSLAM_HIP(hipMalloc(&s->batch_slot_entry, L * sizeof(int)));
    DSLAM_HIP(hipMalloc(&s->batch_marks, L * 64));
    DSLAM_HIP(hipMemsetAsync(s->batch_marks, 0, L * 64, e->stream));   // (every batch leaves them zero again)
    DSLAM_HIP(hipMalloc(&s->batch_order, 8 * L * sizeof(int)));
    DSLAM_HIP(hipMalloc(&s->batch_counters, 16 * sizeof(int)));   // [0..8): blocks per class, [8]: block-operations
    DSLAM_HIP(hipMalloc(&s->batch_ops_dev, 2 * kBatchMax * sizeof(HostBatchOp)));
    DSLAM_HIP(hipMalloc(&s->batch_lists_dev, 3 * kBatchMax * sizeof(HostBatchList)));
    DSLAM_HIP(hipHostMalloc(&s->batch_staging, 2 * kBatchMax * sizeof(HostBatchOp) + 3 * kBatchMax * sizeof(HostBatchList), hipHostMallocDefault));
    DSLAM_HIP(hipEventCreateWithFlags(&s->batch_staging_ev, hipEventDisableTiming));
  }
  if (!fs->batch_lists) {
    DSLAM_HIP(hipMalloc(&fs->batch_lists, fs->list_bytes * kBatchMax));
    DSLAM_HIP(hipMemsetAsync(fs->batch_lists, 0, fs->list_bytes * kBatchMax, e->stream));   // (the count headers)
    fs->batch_list_ptr.resize(kBatchMax);
    for (int i = 0; i < kBatchMax; i++) fs->batch_list_ptr[i] = fs->batch_lists + fs->list_bytes * i;
  }
  (void)e;
  return DSLAM_OK;
}

int dslam_reintegrate_batch(dslam_engine *e, dslam_scene *s, dslam_view *v, dslam_render_state *r, dslam_frame_store *fs,
                            int n, const int32_t *slots, const float *old_M, const float *new_M, const float intr[4],
                            float affine_a, float affine_b) {
  DSLAM_REQUIRE(e && s && v && r && fs && intr && n >= 0 && (n == 0 || (slots && old_M && new_M)), "null argument");
  DSLAM_REQUIRE(s->engine == e && v->engine == e && r->engine == e && fs->engine == e, "objects belong to a different engine");
  DSLAM_REQUIRE(fs->lists, "dslam_frame_store_enable_lists has not been called");
  DSLAM_REQUIRE(v->w_rgb == fs->w_rgb && v->h_rgb == fs->h_rgb && v->w_d == fs->w_d && v->h_d == fs->h_d, "view and frame store sizes differ");
  DSLAM_REQUIRE(fs->list_cap >= r->n_local, "the store's lists are smaller than this render state's visible list");
  DSLAM_REQUIRE(fs->list_entries == s->n_entries && r->n_entries == s->n_entries, "the store's lists / the render state belong to a table of another size");
  if (s->p.use_swapping || s->p.stop_integrating_at_max_w || v->w_rgb != v->w_d || v->h_rgb != v->h_d) {
    set_last_error("dslam_reintegrate_batch: scenes with host swapping or stopIntegratingAtMaxW and views with a separate colour "
                   "camera take the per-keyframe calls (dslam_deprocess_frame_stored + dslam_process_frame)");
    return DSLAM_ERR_UNSUPPORTED;
  }
  for (int k = 0; k < n; k++) {
    DSLAM_REQUIRE(slots[k] >= 0 && slots[k] < fs->capacity, "frame store slot out of range");
    DSLAM_REQUIRE(fs->has_list[slots[k]], "no visible list was stored for a keyframe of the batch");
    for (int j = 0; j < k; j++) DSLAM_REQUIRE(slots[j] != slots[k], "a keyframe appears twice in the batch");
  }
  // everything that can be refused is refused here, before anything changes: what launch_allocate checks per pass
  // (table size above, an invertible pose, the order key's range) for every keyframe of the batch
  for (int k = 0; k < n; k++) {
    float inv[16];
    if (!invert_matrix(new_M + 16 * (size_t)k, inv)) { set_last_error("dslam_reintegrate_batch: a new pose matrix is singular"); return DSLAM_ERR_INVALID; }
  }
  { int cap; const int rc_cap = alloc_step_cap(s, v->w_d, v->h_d, &cap); if (rc_cap) return rc_cap; }
  int rc = batch_scratch(e, s, fs);   // (n = 0: a set-up call -- the buffers exist before the first batch needs them)
  if (rc) return rc;
  if ((rc = ensure_scratch(e, s->n_entries, s->p.num_local_blocks))) return rc;
  if (n == 0) return DSLAM_OK;
  e->view_reads++;  // (see dslam_engine::last_fence)
  s->version = next_map_version();  // the map changes: GetImage memos of this scene are stale
  r->memo_valid = false;
  const size_t L = (size_t)s->p.num_local_blocks;
  const size_t ids_off = kListHeader, pos_off = kListHeader + (size_t)fs->list_cap * sizeof(int);
  for (int first = 0; first < n; first += kBatchMax) {
    const int K = (n - first) < kBatchMax ? (n - first) : kBatchMax;
    DSLAM_HIP(hipMemsetAsync(s->batch_born, 0, L * sizeof(int), e->stream));
    DSLAM_HIP(hipMemsetAsync(s->batch_counters, 0, 16 * sizeof(int), e->stream));
    // (built in page-locked memory: the copies are queued behind the allocation passes and nothing waits for them here)
    DSLAM_HIP(hipEventSynchronize(s->batch_staging_ev));   // the previous batch's copies have left the buffer
    HostBatchOp *ops = reinterpret_cast<HostBatchOp *>(s->batch_staging);
    HostBatchList *lists = reinterpret_cast<HostBatchList *>(ops + 2 * kBatchMax);   // [2K, 3K): the lists whose block positions are filled in at the end
    // phase 1: the allocation passes of the K re-fusions, in keyframe order (they read the table and the keyframes' depth
    // images, not the voxels); every pass' list goes to a scratch buffer, the blocks it allocates are stamped
    s->alloc_born = s->batch_born;
    int n_pos_jobs = 0;
    float *const view_depth = v->depth;
    for (int k = 0; k < K; k++) {
      const int slot = slots[first + k];
      const void *rgba = fs->rgba + fs->rgba_bytes * slot, *raw = fs->depth + fs->depth_bytes * slot;
      if ((rc = launch_view_convert(e, v, rgba, raw, affine_a, affine_b))) break;
      // the pass derives the keyframe's metric depth image (as UpdateView would) -- into the batch's own image k, where
      // both operations of the keyframe read it in the block launch
      v->depth = s->batch_depth + (size_t)k * s->batch_depth_pixels;
      s->alloc_born_stamp = k + 1;
      // the pass writes its list straight into the scratch list of keyframe k; the last one into the render state's own
      // (what the loop of per-keyframe calls leaves there), from where it is copied
      unsigned char *nb = fs->batch_list_ptr[k];
      const bool last_pass = first + k == n - 1;
      if ((rc = launch_allocate(e, s, v, r, new_M + 16 * (size_t)(first + k), intr, 0, last_pass ? nullptr : reinterpret_cast<int *>(nb + ids_off),
                                last_pass ? nullptr : nb)))
        break;
      int bit = 0, frame = 0;
      if ((rc = prepare_push_visible_list(e, s, 1, &bit, &frame))) break;   // (ProcessFrame(isDefusion) queues on ring 1)
      if (last_pass) {
        if ((rc = launch_store_visible_list(e, s, r, nb, reinterpret_cast<int *>(nb + ids_off), reinterpret_cast<short4 *>(nb + pos_off),
                                            fs->list_cap)))
          break;
      } else {
        lists[2 * K + n_pos_jobs++] = {nb, nb + ids_off, nb + pos_off};
      }
      const unsigned char *ob = list_slot(fs, slot);
      HostBatchOp &d = ops[2 * k], &f = ops[2 * k + 1];
      memcpy(d.M, old_M + 16 * (size_t)(first + k), 64);
      memcpy(f.M, new_M + 16 * (size_t)(first + k), 64);
      d.depth = f.depth = v->depth; d.rgba = f.rgba = rgba;
      d.push_bit = d.push_frame = 0; f.push_bit = bit; f.push_frame = frame;
      lists[2 * k] = {ob, ob + ids_off, ob + pos_off};
      lists[2 * k + 1] = {nb, nb + ids_off, nullptr};
    }
    s->alloc_born = nullptr;
    v->depth = view_depth;
    v->depth_dirty = true;   // (the view's own float image was not written: its next consumer derives it)
    if (rc) return batch_failed(s, r, fs, slots + first, K, rc);
    // phase 2: which operations touch which block, then every touched block once
    DSLAM_HIP(hipMemcpyAsync(s->batch_ops_dev, ops, 2 * (size_t)K * sizeof(HostBatchOp), hipMemcpyHostToDevice, e->stream));
    DSLAM_HIP(hipMemcpyAsync(s->batch_lists_dev, lists, 3 * (size_t)K * sizeof(HostBatchList), hipMemcpyHostToDevice, e->stream));
    DSLAM_HIP(hipEventRecord(s->batch_staging_ev, e->stream));
    if ((rc = launch_batch_ops(e, s->batch_lists_dev, 2 * K, s, s->batch_born, s->batch_marks, s->batch_opmask, s->batch_slot_entry,
                               s->batch_order, s->batch_counters)))
      return batch_failed(s, r, fs, slots + first, K, rc);
    if ((rc = launch_reintegrate_blocks(e, s, v->w_d, v->h_d, v->w_rgb, v->h_rgb, intr, s->batch_ops_dev,
                                        s->batch_opmask, s->batch_slot_entry, s->batch_order, s->batch_counters, 1, 2 * K)))
      return batch_failed(s, r, fs, slots + first, K, rc);
    if ((rc = launch_store_list_positions(e, s, reinterpret_cast<const HostBatchList *>(s->batch_lists_dev) + 2 * K, n_pos_jobs)))
      return batch_failed(s, r, fs, slots + first, K, rc);
    // the lists of the re-fusions become the keyframes' stored lists: the buffers trade places
    for (int k = 0; k < K; k++) std::swap(fs->list_ptr[slots[first + k]], fs->batch_list_ptr[k]);
  }
  return finish_call(e);
}

int dslam_reintegrate_batch_stats(dslam_engine *e, const dslam_scene *s, int32_t *blocks_out, int32_t *block_operations_out) {
  DSLAM_REQUIRE(e && s && s->engine == e, "null argument");
  DSLAM_REQUIRE(s->batch_counters, "no batch has run on this scene");
  int host[16];
  DSLAM_HIP(hipMemcpyAsync(host, s->batch_counters, sizeof(host), hipMemcpyDeviceToHost, e->stream));
  const int rc = sync_check(e);
  if (rc) return rc;
  int blocks = 0;
  for (int c = 0; c < 8; c++) blocks += host[c];
  if (blocks_out) *blocks_out = blocks;
  if (block_operations_out) *block_operations_out = host[8];
  return DSLAM_OK;
}

// ---- depthPostProcessing -------------------------------------------------------------------------------------
static int depth_post_args(dslam_engine *e, const void *curr, const void *prev, int w, int h, const float *Tpc,
                           const float *intr) {
  DSLAM_REQUIRE(e && curr && prev && Tpc && intr, "null argument");
  DSLAM_REQUIRE(w > 0 && h > 0, "bad image size");
  return DSLAM_OK;
}

int dslam_depth_post_processing_device(dslam_engine *e, void *curr_dev, const void *prev_dev, int w, int h,
                                       const float Tpc[16], const float intr[4], float threshold, float area,
                                       int *count_out) {
  int rc = depth_post_args(e, curr_dev, prev_dev, w, h, Tpc, intr);
  if (rc) return rc;
  int *count_dev = e->misc_counter;
  if ((rc = launch_depth_post(e, (short *)curr_dev, (const unsigned short *)prev_dev, w, h, Tpc, intr, threshold, area, count_dev))) return rc;
  if (count_out) {
    DSLAM_HIP(hipMemcpyAsync(e->pinned, count_dev, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    DSLAM_HIP(hipStreamSynchronize(e->stream));
    *count_out = *(int *)e->pinned;
    return DSLAM_OK;
  }
  return finish_call(e);
}

int dslam_depth_post_processing(dslam_engine *e, int16_t *curr_host, const int16_t *prev_host, int w, int h,
                                const float Tpc[16], const float intr[4], float threshold, float area,
                                int *count_out) {
  int rc = depth_post_args(e, curr_host, prev_host, w, h, Tpc, intr);
  if (rc) return rc;
  const size_t d_bytes = (size_t)w * h * 2;
  if ((rc = ensure_staging(e, 2 * d_bytes))) return rc;
  DSLAM_HIP(hipStreamSynchronize(e->stream));
  memcpy(e->staging_host, curr_host, d_bytes);
  memcpy((char *)e->staging_host + d_bytes, prev_host, d_bytes);
  DSLAM_HIP(hipMemcpyAsync(e->staging_dev, e->staging_host, 2 * d_bytes, hipMemcpyHostToDevice, e->stream));
  int *count_dev = e->misc_counter;
  if ((rc = launch_depth_post(e, (short *)e->staging_dev, (const unsigned short *)((char *)e->staging_dev + d_bytes), w, h, Tpc,
                              intr, threshold, area, count_dev)))
    return rc;
  DSLAM_HIP(hipMemcpyAsync(e->staging_host, e->staging_dev, d_bytes, hipMemcpyDeviceToHost, e->stream));
  DSLAM_HIP(hipMemcpyAsync(e->pinned, count_dev, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  DSLAM_HIP(hipStreamSynchronize(e->stream));
  memcpy(curr_host, e->staging_host, d_bytes);
  if (count_out) *count_out = *(int *)e->pinned;
  return DSLAM_OK;
}

// ---- fusion --------------------------------------------------------------------------------------------------
int dslam_set_fusion_weight_params(dslam_engine *e, const dslam_weight_params *w) {
  DSLAM_REQUIRE(e && w, "null argument");
  // newW <= 255: the voxel weight is one byte and k_integrate indexes its reciprocal table with w_depth + newW <= 510
  DSLAM_REQUIRE(!w->depth_weighting || (w->max_distance > 0 && w->max_new_w >= 1 && w->max_new_w <= 255),
                "bad weight params (need max_distance > 0 and 1 <= max_new_w <= 255)");
  e->wp = *w;
  return DSLAM_OK;
}

static int check_frame_args(dslam_engine *e, dslam_scene *s, const dslam_view *v, const dslam_render_state *r,
                            const float *M, const float *intr) {
  DSLAM_REQUIRE(e && s && v && r && M && intr, "null argument");
  DSLAM_REQUIRE(s->engine == e && v->engine == e && r->engine == e, "objects belong to a different engine");
  e->view_reads++;  // (see dslam_engine::last_fence)
  return DSLAM_OK;
}

int dslam_allocate_scene_from_depth(dslam_engine *e, dslam_scene *s, const dslam_view *v, dslam_render_state *r,
                                    const float M_d[16], const float intr[4], int only_visible) {
  int rc = check_frame_args(e, s, v, r, M_d, intr);
  if (rc) return rc;
  s->version = next_map_version();  // the map may change: GetImage memos of this scene are stale
  if (r) r->memo_valid = false;
  rc = launch_allocate(e, s, v, r, M_d, intr, only_visible);
  if (rc) return rc;
  return finish_call(e);
}

int dslam_integrate_into_scene(dslam_engine *e, dslam_scene *s, const dslam_view *v, const dslam_render_state *r,
                               const float M_d[16], const float intr_d[4], const float M_rgb[16],
                               const float intr_rgb[4]) {
  int rc = check_frame_args(e, s, v, r, M_d, intr_d);
  if (rc) return rc;
  s->version = next_map_version();  // the map may change: GetImage memos of this scene are stale
  rc = launch_integrate(e, s, v, r, M_d, intr_d, M_rgb, intr_rgb, false);
  if (rc) return rc;
  return finish_call(e);
}

int dslam_process_frame(dslam_engine *e, dslam_scene *s, const dslam_view *v, dslam_render_state *r,
                        const float M_d[16], const float intr_d[4], const float M_rgb[16], const float intr_rgb[4],
                        int only_visible, int is_defusion) {
  int rc = check_frame_args(e, s, v, r, M_d, intr_d);
  if (rc) return rc;
  s->version = next_map_version();  // the map may change: GetImage memos of this scene are stale
  if (r) r->memo_valid = false;
  if ((rc = launch_allocate(e, s, v, r, M_d, intr_d, only_visible))) return rc;
  if ((rc = launch_integrate(e, s, v, r, M_d, intr_d, M_rgb, intr_rgb, false, is_defusion ? 1 : 0))) return rc;
  if (s->p.use_swapping) {
    if ((rc = launch_swap_in(e, s, r))) return rc;
    if ((rc = launch_swap_out(e, s, r, false))) return rc;
  }
  return finish_call(e);
}

int dslam_deprocess_frame(dslam_engine *e, dslam_scene *s, const dslam_view *v, dslam_render_state *r,
                          const float M_d[16], const float intr_d[4], const float M_rgb[16], const float intr_rgb[4]) {
  int rc = check_frame_args(e, s, v, r, M_d, intr_d);
  if (rc) return rc;
  s->version = next_map_version();  // the map may change: GetImage memos of this scene are stale
  if (r) r->memo_valid = false;
  if ((rc = launch_allocate(e, s, v, r, M_d, intr_d, 1))) return rc;
  if ((rc = launch_integrate(e, s, v, r, M_d, intr_d, M_rgb, intr_rgb, true))) return rc;
  return finish_call(e);
}

// ---- depth tracker -------------------------------------------------------------------------------------------
int dslam_track_camera(dslam_engine *e, const dslam_view *v, dslam_render_state *r, const float scene_pose_M[16],
                       float pose_M[16], const float intr[4], const dslam_tracker_params *params,
                       dslam_tracker_result *result) {
  DSLAM_REQUIRE(e && v && r && scene_pose_M && pose_M && intr && params, "null argument");
  e->view_reads++;  // (see dslam_engine::last_fence)
  DSLAM_REQUIRE(v->engine == e && r->engine == e, "objects belong to a different engine");
  return launch_track_camera(e, v, r, scene_pose_M, pose_M, intr, params, result);
}

// ---- decay / sliding window ----------------------------------------------------------------------------------
int dslam_decay(dslam_engine *e, dslam_scene *s, dslam_render_state *r, int max_weight, int min_age, int force_all) {
  DSLAM_REQUIRE(e && s, "null argument");
  s->version = next_map_version();  // the map may change: GetImage memos of this scene are stale
  if (r) r->memo_valid = false;
  int rc = launch_decay(e, s, r, max_weight, min_age, force_all, 0);
  if (rc) return rc;
  return finish_call(e);
}
int dslam_decay_defusion_part(dslam_engine *e, dslam_scene *s, dslam_render_state *r, int max_weight, int min_age,
                              int force_all) {
  DSLAM_REQUIRE(e && s, "null argument");
  s->version = next_map_version();  // the map may change: GetImage memos of this scene are stale
  if (r) r->memo_valid = false;
  int rc = launch_decay(e, s, r, max_weight, min_age, force_all, 1);
  if (rc) return rc;
  return finish_call(e);
}
int dslam_slide_window(dslam_engine *e, dslam_scene *s, dslam_render_state *r, int max_age) {
  DSLAM_REQUIRE(e && s, "null argument");
  s->version = next_map_version();  // the map may change: GetImage memos of this scene are stale
  if (r) r->memo_valid = false;
  if (max_age < 0) max_age = 0;
  while (s->ring_next[0] - s->ring_head[0] > max_age) {
    int rc = launch_slide_pop(e, s, r, 0);
    if (rc) return rc;
  }
  return finish_call(e);
}
int dslam_slide_window_defusion_part(dslam_engine *e, dslam_scene *s, dslam_render_state *r, int max_age, int max_size) {
  DSLAM_REQUIRE(e && s, "null argument");
  s->version = next_map_version();  // the map may change: GetImage memos of this scene are stale
  if (r) r->memo_valid = false;
  (void)max_age;
  if (max_size < 0) max_size = 0;
  while (s->ring_next[1] - s->ring_head[1] > max_size) {
    int rc = launch_slide_pop(e, s, r, 1);
    if (rc) return rc;
  }
  return finish_call(e);
}

// ---- swapping ------------------------------------------------------------------------------------------------
int dslam_swap_in(dslam_engine *e, dslam_scene *s, dslam_render_state *r) {
  DSLAM_REQUIRE(e && s && s->p.use_swapping, "scene was created without swapping");
  s->version = next_map_version();  // the map may change: GetImage memos of this scene are stale
  if (r) r->memo_valid = false;
  int rc = launch_swap_in(e, s, r);
  if (rc) return rc;
  return finish_call(e);
}
int dslam_swap_out(dslam_engine *e, dslam_scene *s, dslam_render_state *r) {
  DSLAM_REQUIRE(e && s && r && s->p.use_swapping, "scene was created without swapping");
  s->version = next_map_version();  // the map may change: GetImage memos of this scene are stale
  if (r) r->memo_valid = false;
  int rc = launch_swap_out(e, s, r, false);
  if (rc) return rc;
  return finish_call(e);
}
int dslam_save_to_global_memory(dslam_engine *e, dslam_scene *s) {
  DSLAM_REQUIRE(e && s && s->p.use_swapping, "scene was created without swapping");
  s->version = next_map_version();  // the map may change: GetImage memos of this scene are stale
  int rc = launch_save_to_global(e, s);
  if (rc) return rc;
  return finish_call(e);
}

// ---- visualisation -------------------------------------------------------------------------------------------
int dslam_find_visible_blocks(dslam_engine *e, const dslam_scene *s, dslam_render_state *r, const float M[16],
                              const float intr[4]) {
  DSLAM_REQUIRE(e && s && r && M && intr, "null argument");
  r->memo_valid = false;  // raycastResult / the lists behind it are about to be rewritten
  r->types_follow_list = false;
  int rc = launch_find_visible(e, s, r, M, intr);
  if (rc) return rc;
  return finish_call(e);
}
int dslam_count_visible_blocks(dslam_engine *e, const dslam_scene *s, const dslam_render_state *r, int min_id,
                               int max_id, int *out) {
  DSLAM_REQUIRE(e && s && r && out, "null argument");
  return launch_count_visible(e, s, r, min_id, max_id, out);
}
int dslam_create_expected_depths(dslam_engine *e, const dslam_scene *s, dslam_render_state *r, const float M[16],
                                 const float intr[4]) {
  DSLAM_REQUIRE(e && s && r && M && intr, "null argument");
  r->memo_valid = false;  // raycastResult / the lists behind it are about to be rewritten
  int rc = launch_expected_depths(e, s, r, M, intr);
  if (rc) return rc;
  return finish_call(e);
}

static int image_out(dslam_engine *e, dslam_render_state *r, int type, uint8_t *out_rgba, float *out_float) {
  const size_t npix = (size_t)r->w * r->h;
  if (out_rgba || out_float) {
    DSLAM_REQUIRE(!(out_rgba && out_float), "pass only one output buffer");
    if (type == DSLAM_IMAGE_DEPTH) {
      DSLAM_REQUIRE(out_float, "DSLAM_IMAGE_DEPTH renders into the float output");
      DSLAM_HIP(hipMemcpyAsync(out_float, r->image_float, npix * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    } else {
      DSLAM_REQUIRE(out_rgba, "this image type renders into the rgba output");
      DSLAM_HIP(hipMemcpyAsync(out_rgba, r->image_rgba, npix * 4, hipMemcpyDeviceToHost, e->stream));
    }
    return sync_check(e);  // host buffers are valid on return
  }
  return finish_call(e);
}

int dslam_render_image(dslam_engine *e, const dslam_scene *s, dslam_render_state *r, const float M[16],
                       const float intr[4], int type, uint8_t *out_rgba, float *out_float) {
  DSLAM_REQUIRE(e && s && r && M && intr, "null argument");
  r->memo_valid = false;  // raycastResult / the lists behind it are about to be rewritten
  DSLAM_REQUIRE(type >= 0 && type <= DSLAM_IMAGE_DEPTH, "unknown image type");
  int rc = launch_render(e, s, r, M, intr, type);
  if (rc) return rc;
  return image_out(e, r, type, out_rgba, out_float);
}

// FindVisibleBlocks + CreateExpectedDepths + march for (scene version, pose, intrinsics), unless the render state
// still holds exactly that (see dslam_render_state::memo_*); then the image of the requested type.
static int get_image_on_device(dslam_engine *e, const dslam_scene *s, dslam_render_state *r, const float *M,
                               const float *intr, int type, void *direct_out = nullptr) {
  // a map whose voxel blocks live in caller memory can change behind the engine's back: never memoised
  const bool hit = r->memo_valid && !s->voxels_external && r->memo_scene == s && r->memo_version == s->version &&
                   r->memo_budget == e->render_tile_budget && memcmp(r->memo_M, M, sizeof(r->memo_M)) == 0 &&
                   memcmp(r->memo_intr, intr, sizeof(r->memo_intr)) == 0;
  int rc;
  if (hit) return launch_render(e, s, r, M, intr, type, true, direct_out);
  r->memo_valid = false;
  r->types_follow_list = false;  // FindVisibleBlocks replaces this render state's list
  if ((rc = launch_find_visible_and_depths(e, s, r, M, intr))) return rc;
  if ((rc = launch_render(e, s, r, M, intr, type, false, direct_out))) return rc;
  r->memo_valid = true; r->memo_scene = s; r->memo_version = s->version; r->memo_budget = e->render_tile_budget;
  memcpy(r->memo_M, M, sizeof(r->memo_M));
  memcpy(r->memo_intr, intr, sizeof(r->memo_intr));
  return DSLAM_OK;
}

int dslam_get_image(dslam_engine *e, const dslam_scene *s, dslam_render_state *r, const float M[16],
                    const float intr[4], int type, uint8_t *out_rgba, float *out_float) {
  DSLAM_REQUIRE(e && s && r && M && intr, "null argument");
  DSLAM_REQUIRE(type >= 0 && type <= DSLAM_IMAGE_DEPTH, "unknown image type");
  // a page-locked output image (dslam_host_alloc) is written by the render kernel itself: no copy behind the kernel
  void *out = type == DSLAM_IMAGE_DEPTH ? (void *)out_float : (void *)out_rgba;
  const size_t bytes = (size_t)r->w * r->h * 4;
  if (out && !(out_rgba && out_float) && in_pinned_range(out, bytes)) {
    int rc = get_image_on_device(e, s, r, M, intr, type, out);
    if (rc) return rc;
    // synchronous engine: the image is there on return (what the reference's callers assume); async engine: the
    // caller pipelines and learns from a fence (dslam_fence_*) when the kernel has stored the last pixel
    return finish_call(e);
  }
  int rc = get_image_on_device(e, s, r, M, intr, type);
  if (rc) return rc;
  return image_out(e, r, type, out_rgba, out_float);
}

int dslam_get_depth_image_int16(dslam_engine *e, const dslam_scene *s, dslam_render_state *r, const float M[16],
                                const float intr[4], int scale, int16_t *out_host) {
  DSLAM_REQUIRE(e && s && r && M && intr && out_host && scale > 0, "bad argument");
  int rc;
  if ((rc = get_image_on_device(e, s, r, M, intr, DSLAM_IMAGE_DEPTH))) return rc;
  // the RGBA image buffer of the render state is idle in this mode: it takes the int16 image (2 of its 4 bytes per pixel)
  const int n = r->w * r->h;
  short *tmp = reinterpret_cast<short *>(r->image_rgba);
  if ((rc = launch_depth_to_int16(e, r->image_float, tmp, n, scale))) return rc;
  DSLAM_HIP(hipMemcpyAsync(out_host, tmp, (size_t)n * 2, hipMemcpyDeviceToHost, e->stream));
  return sync_check(e);
}

int dslam_create_icp_maps(dslam_engine *e, const dslam_scene *s, dslam_render_state *r, const float M[16],
                          const float intr[4], float *out_points, float *out_normals) {
  DSLAM_REQUIRE(e && s && r && M && intr, "null argument");
  r->memo_valid = false;  // raycastResult / the lists behind it are about to be rewritten
  int rc;
  if ((rc = launch_expected_depths(e, s, r, M, intr))) return rc;
  if ((rc = launch_icp_maps(e, s, r, M, intr))) return rc;
  const size_t bytes = (size_t)r->w * r->h * sizeof(float4);
  if (out_points) DSLAM_HIP(hipMemcpyAsync(out_points, r->icp_points, bytes, hipMemcpyDeviceToHost, e->stream));
  if (out_normals) DSLAM_HIP(hipMemcpyAsync(out_normals, r->icp_normals, bytes, hipMemcpyDeviceToHost, e->stream));
  if (out_points || out_normals) return sync_check(e);
  return finish_call(e);
}

int dslam_download_raycast_image(dslam_engine *e, const dslam_render_state *r, uint8_t *out_rgba) {
  DSLAM_REQUIRE(e && r && out_rgba, "null argument");
  DSLAM_REQUIRE(r->raycast_image, "dslam_create_icp_maps has not run on this render state");
  DSLAM_HIP(hipMemcpyAsync(out_rgba, r->raycast_image, (size_t)r->w * r->h * 4, hipMemcpyDeviceToHost, e->stream));
  return sync_check(e);
}

int dslam_download_icp_maps(dslam_engine *e, const dslam_render_state *r, float *out_points, float *out_normals) {
  DSLAM_REQUIRE(e && r, "null argument");
  DSLAM_REQUIRE(r->icp_points && r->icp_normals, "dslam_create_icp_maps has not run on this render state");
  const size_t bytes = (size_t)r->w * r->h * sizeof(float4);
  if (out_points) DSLAM_HIP(hipMemcpyAsync(out_points, r->icp_points, bytes, hipMemcpyDeviceToHost, e->stream));
  if (out_normals) DSLAM_HIP(hipMemcpyAsync(out_normals, r->icp_normals, bytes, hipMemcpyDeviceToHost, e->stream));
  return sync_check(e);
}

// ---- meshing export --------------------------------------------------------------------------------------------
int dslam_mesh_scene(dslam_engine *e, const dslam_scene *s, int max_triangles, int with_colour, int *out_num_triangles) {
  DSLAM_REQUIRE(e && s && out_num_triangles, "null argument");
  if (max_triangles <= 0) max_triangles = s->p.num_local_blocks * 32;  // ITMMesh::noMaxTriangles
  int rc = launch_mesh_scene(e, s, max_triangles, with_colour, out_num_triangles);
  if (rc) return rc;
  return finish_call(e);
}

int dslam_mesh_download(dslam_engine *e, float *out_positions, float *out_colours, int capacity_triangles) {
  DSLAM_REQUIRE(e && out_positions && capacity_triangles >= 0, "bad argument");
  DSLAM_REQUIRE(capacity_triangles >= e->mesh_triangles, "dslam_mesh_download: buffer smaller than the mesh");
  DSLAM_REQUIRE(!out_colours || e->mesh_has_colour, "dslam_mesh_download: the mesh was made without colours");
  const size_t bytes = (size_t)e->mesh_triangles * 9 * sizeof(float);
  if (bytes) {
    DSLAM_HIP(hipMemcpyAsync(out_positions, e->mesh_positions, bytes, hipMemcpyDeviceToHost, e->stream));
    if (out_colours) DSLAM_HIP(hipMemcpyAsync(out_colours, e->mesh_colours, bytes, hipMemcpyDeviceToHost, e->stream));
  }
  return sync_check(e);
}

// ---- read-back -----------------------------------------------------------------------------------------------
int dslam_get_stats(dslam_engine *e, const dslam_scene *s, const dslam_render_state *r, dslam_stats *out) {
  DSLAM_REQUIRE(e && s && out, "null argument");
  char *host = reinterpret_cast<char *>(e->pinned);
  SceneCounters *sc = reinterpret_cast<SceneCounters *>(host);
  RenderCounters *rc = reinterpret_cast<RenderCounters *>(host + 128);
  DSLAM_HIP(hipMemcpyAsync(sc, s->counters, sizeof(SceneCounters), hipMemcpyDeviceToHost, e->stream));
  if (r) DSLAM_HIP(hipMemcpyAsync(rc, r->counters, sizeof(RenderCounters), hipMemcpyDeviceToHost, e->stream));
  const int rc_engine = sync_check(e);   // (what any scene of the engine reported since the last synchronising call)
  memset(out, 0, sizeof(*out));
  out->num_allocated_blocks = s->p.num_local_blocks;
  out->last_free_block_id = sc->last_free;
  out->last_free_excess_id = sc->last_free_ex;
  out->no_visible_entries = r ? rc->no_visible : 0;
  out->decayed_block_count = sc->decayed_blocks;
  out->slid_block_count = sc->slid_blocks;
  out->frame_counter = s->frame_counter;
  out->fusion_fifo_len = s->ring_next[0] - s->ring_head[0];
  out->defusion_fifo_len = s->ring_next[1] - s->ring_head[1];
  out->alloc_failures = sc->alloc_failures;
  out->last_swapped_in = sc->swapped_in;
  out->last_swapped_out = sc->swapped_out;
  if (sc->error_flags & 2) {
    set_last_error("a tile count of an ordered compaction never arrived (device made no progress?): the map state is undefined");
    return DSLAM_ERR_HIP;
  }
  if (sc->error_flags & 1) {
    set_last_error("allocation ray needed more steps than the order key encodes (non-rigid pose or mu/voxel_size changed?)");
    return DSLAM_ERR_UNSUPPORTED;
  }
  return rc_engine;
}

static int d2h(dslam_engine *e, void *dst, const void *src, size_t bytes) {
  DSLAM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, e->stream));
  return sync_check(e);
}
static int h2d(dslam_engine *e, void *dst, const void *src, size_t bytes) {
  DSLAM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, e->stream));
  DSLAM_HIP(hipStreamSynchronize(e->stream));
  return DSLAM_OK;
}

int dslam_download_hash_table(dslam_engine *e, const dslam_scene *s, dslam_hash_entry *out) {
  DSLAM_REQUIRE(e && s && out, "null argument");
  return d2h(e, out, s->hash, (size_t)s->n_entries * sizeof(HashEntry));
}
int dslam_download_voxel_blocks(dslam_engine *e, const dslam_scene *s, int first, int n, dslam_voxel *out) {
  DSLAM_REQUIRE(e && s && out && first >= 0 && n >= 0 && first + n <= s->p.num_local_blocks, "bad block range");
  return d2h(e, out, s->voxels + (size_t)first * kBlock3, (size_t)n * kBlock3 * sizeof(uint2));
}
int dslam_download_allocation_list(dslam_engine *e, const dslam_scene *s, int32_t *out) {
  DSLAM_REQUIRE(e && s && out, "null argument");
  return d2h(e, out, s->alloc_list, (size_t)s->p.num_local_blocks * sizeof(int));
}
int dslam_download_excess_list(dslam_engine *e, const dslam_scene *s, int32_t *out) {
  DSLAM_REQUIRE(e && s && out, "null argument");
  return d2h(e, out, s->excess_list, (size_t)s->p.num_excess * sizeof(int));
}
int dslam_download_visible_ids(dslam_engine *e, const dslam_render_state *r, int32_t *out, int capacity, int *count) {
  DSLAM_REQUIRE(e && r && out, "null argument");
  RenderCounters *rc = reinterpret_cast<RenderCounters *>(reinterpret_cast<char *>(e->pinned) + 128);
  int st = d2h(e, rc, r->counters, sizeof(RenderCounters));
  if (st) return st;
  int n = rc->no_visible < capacity ? rc->no_visible : capacity;
  if (count) *count = rc->no_visible;
  if (n > 0) return d2h(e, out, r->visible_ids, (size_t)n * sizeof(int));
  return DSLAM_OK;
}
int dslam_download_visible_types(dslam_engine *e, const dslam_render_state *r, uint8_t *out) {
  DSLAM_REQUIRE(e && r && out, "null argument");
  int rc = d2h(e, out, r->visible_type, r->n_entries);
  for (int i = 0; i < r->n_entries; i++) out[i] &= 0x7f;  // the device bytes carry the pass' generation bit
  return rc;
}
int dslam_download_range_image(dslam_engine *e, const dslam_render_state *r, float *out) {
  DSLAM_REQUIRE(e && r && out, "null argument");
  return d2h(e, out, r->range, (size_t)r->w * r->h * sizeof(float2));
}
int dslam_download_raycast_result(dslam_engine *e, const dslam_render_state *r, float *out) {
  DSLAM_REQUIRE(e && r && out, "null argument");
  return d2h(e, out, r->raycast, (size_t)r->w * r->h * sizeof(float4));
}
int dslam_download_view_depth(dslam_engine *e, const dslam_view *v, float *out) {
  DSLAM_REQUIRE(e && v && out, "null argument");
  e->view_reads++;  // (see dslam_engine::last_fence)
  int rc = ensure_view_depth(e, v);
  if (rc) return rc;
  return d2h(e, out, v->depth, (size_t)v->w_d * v->h_d * sizeof(float));
}
int dslam_download_swap_states(dslam_engine *e, const dslam_scene *s, uint8_t *out) {
  DSLAM_REQUIRE(e && s && out && s->swap_state, "scene has no swap state");
  return d2h(e, out, s->swap_state, s->n_entries);
}
int dslam_download_last_seen(dslam_engine *e, const dslam_scene *s, int32_t *out) {
  DSLAM_REQUIRE(e && s && out, "null argument");
  return d2h(e, out, s->last_seen, (size_t)s->p.num_local_blocks * sizeof(int));
}
int dslam_download_stored_block(dslam_engine *e, const dslam_scene *s, int entry, dslam_voxel *out) {
  DSLAM_REQUIRE(e && s && s->slot_dev && entry >= 0 && entry < s->n_entries, "bad entry / no global cache");
  int slot = -1;  // (d2h waits for the stream: the swap kernels write the host slabs directly)
  int rc = d2h(e, &slot, s->slot_dev + entry, sizeof(int));
  if (rc) return rc;
  if (out) {
    if (slot >= 0) memcpy(out, s->slabs[slot >> kSlabShift] + (size_t)(slot & (kSlabBlocks - 1)) * (kBlock3 / 2), kBlock3 * sizeof(dslam_voxel));
    else memset(out, 0, kBlock3 * sizeof(dslam_voxel));
  }
  return slot >= 0 ? 1 : 0;
}
int dslam_download_alloc_scratch(dslam_engine *e, const dslam_scene *s, uint8_t *types, int16_t *coords) {
  DSLAM_REQUIRE(e && s && e->scratch_entries >= s->n_entries, "no allocation pass has run");
  int rc = 0;
  if (types) rc = d2h(e, types, e->alloc_type, s->n_entries);
  if (!rc && coords) rc = d2h(e, coords, e->block_coords, (size_t)s->n_entries * sizeof(short4));
  return rc;
}

int dslam_upload_scene_state(dslam_engine *e, dslam_scene *s, const dslam_hash_entry *hash, const int32_t *alloc_list,
                             int last_free, const int32_t *excess_list, int last_free_ex) {
  DSLAM_REQUIRE(e && s, "null argument");
  s->version = next_map_version();  // the map may change: GetImage memos of this scene are stale
  int rc = 0;
  if (hash) {
    rc = h2d(e, s->hash, hash, (size_t)s->n_entries * sizeof(HashEntry));
    if (!rc) rc = launch_build_alloc_bits(e, s);
    if (!rc) rc = finish_call(e);
  }
  SceneCounters *sc = reinterpret_cast<SceneCounters *>(e->pinned);
  if (!rc && (alloc_list || excess_list)) {
    rc = d2h(e, sc, s->counters, sizeof(SceneCounters));
    if (!rc && alloc_list) {
      rc = h2d(e, s->alloc_list, alloc_list, (size_t)s->p.num_local_blocks * sizeof(int));
      sc->last_free = last_free;
    }
    if (!rc && excess_list) {
      rc = h2d(e, s->excess_list, excess_list, (size_t)s->p.num_excess * sizeof(int));
      sc->last_free_ex = last_free_ex;
    }
    if (!rc) rc = h2d(e, s->counters, sc, sizeof(SceneCounters));
  }
  return rc;
}
int dslam_upload_voxel_blocks(dslam_engine *e, dslam_scene *s, int first, int n, const dslam_voxel *host) {
  DSLAM_REQUIRE(e && s && host && first >= 0 && n >= 0 && first + n <= s->p.num_local_blocks, "bad block range");
  s->version = next_map_version();  // the map may change: GetImage memos of this scene are stale
  return h2d(e, s->voxels + (size_t)first * kBlock3, host, (size_t)n * kBlock3 * sizeof(uint2));
}
int dslam_upload_visible_ids(dslam_engine *e, dslam_render_state *r, const int32_t *ids, int count) {
  DSLAM_REQUIRE(e && r && ids && count >= 0 && count <= r->n_local, "bad visible list");
  r->memo_valid = false;  // raycastResult / the lists behind it are about to be rewritten
  r->types_follow_list = false;
  int rc = h2d(e, r->visible_ids, ids, (size_t)count * sizeof(int));
  if (rc) return rc;
  RenderCounters *rcn = reinterpret_cast<RenderCounters *>(reinterpret_cast<char *>(e->pinned) + 128);
  rc = d2h(e, rcn, r->counters, sizeof(RenderCounters));
  if (rc) return rc;
  rcn->no_visible = count;
  __atomic_store_n(r->vis_hint, count, __ATOMIC_RELAXED);
  return h2d(e, r->counters, rcn, sizeof(RenderCounters));
}

void *dslam_scene_voxel_blocks_dev(dslam_scene *s) { return s ? s->voxels : nullptr; }
void *dslam_scene_hash_table_dev(dslam_scene *s) { return s ? s->hash : nullptr; }
int dslam_scene_table_changed(dslam_engine *e, dslam_scene *s) {
  DSLAM_REQUIRE(e && s && s->engine == e, "null argument");
  s->version = next_map_version();  // the map changed: GetImage memos of this scene are stale
  int rc = launch_build_alloc_bits(e, s);
  if (rc) return rc;
  return finish_call(e);
}
void *dslam_render_state_image_dev(dslam_render_state *r, int want_float) {
  if (!r) return nullptr;
  return want_float ? (void *)r->image_float : (void *)r->image_rgba;
}

// ---- instrumentation -----------------------------------------------------------------------------------------
int dslam_time_integrate(dslam_engine *e, dslam_scene *s, const dslam_view *v, const dslam_render_state *r,
                         const float M_d[16], const float intr[4], int iterations, float *out_ms, int *out_blocks) {
  int rc = check_frame_args(e, s, v, r, M_d, intr);
  if (rc) return rc;
  s->version = next_map_version();  // the map may change: GetImage memos of this scene are stale
  DSLAM_REQUIRE(iterations > 0 && out_ms, "bad argument");
  e->view_reads++;  // (see dslam_engine::last_fence)
  hipEvent_t a, b;
  DSLAM_HIP(hipEventCreate(&a));
  DSLAM_HIP(hipEventCreate(&b));
  const bool saved = e->timer_enabled;
  e->timer_enabled = false;
  rc = launch_integrate(e, s, v, r, M_d, intr, nullptr, nullptr, false);  // warm-up
  DSLAM_HIP(hipEventRecord(a, e->stream));
  for (int i = 0; i < iterations && !rc; i++) rc = launch_integrate(e, s, v, r, M_d, intr, nullptr, nullptr, false);
  DSLAM_HIP(hipEventRecord(b, e->stream));
  DSLAM_HIP(hipEventSynchronize(b));
  e->timer_enabled = saved;
  float ms = 0;
  DSLAM_HIP(hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  *out_ms = ms / iterations;
  if (out_blocks) {
    RenderCounters *rcn = reinterpret_cast<RenderCounters *>(reinterpret_cast<char *>(e->pinned) + 128);
    int st = d2h(e, rcn, r->counters, sizeof(RenderCounters));
    if (st) return st;
    *out_blocks = rcn->no_visible;
  }
  return rc;
}

int dslam_kernel_timer_enable(dslam_engine *e, int enable) {
  DSLAM_REQUIRE(e, "null engine");
  DSLAM_HIP(hipStreamSynchronize(e->stream));
  if (enable && e->ev_pool.empty()) {
    const size_t n = 2 * 8192;  // up to 8192 timed launches between reads
    DSLAM_REQUIRE(e->pinned_bytes >= 256 + 8192 * sizeof(int), "pinned mirror too small");
    DSLAM_HIP(hipMalloc(&e->timer_counts_dev, 8192 * sizeof(int)));
    e->ev_pool.resize(n);
    for (size_t i = 0; i < n; i++) DSLAM_HIP(hipEventCreate(&e->ev_pool[i]));
  }
  e->timer_enabled = enable != 0;
  e->ev_used = 0;
  e->timer_ms = 0; e->timer_launches = 0; e->timer_blocks = 0;
  return DSLAM_OK;
}

int dslam_kernel_timer_read(dslam_engine *e, double *out_ms, int64_t *out_launches, int64_t *out_blocks) {
  DSLAM_REQUIRE(e, "null engine");
  DSLAM_HIP(hipStreamSynchronize(e->stream));
  int *counts = reinterpret_cast<int *>(e->pinned) + 64;
  if (e->ev_used > 0) {
    DSLAM_HIP(hipMemcpyAsync(counts, e->timer_counts_dev, (e->ev_used / 2) * sizeof(int), hipMemcpyDeviceToHost, e->stream));
    DSLAM_HIP(hipStreamSynchronize(e->stream));
  }
  for (size_t i = 0; i + 1 < e->ev_used; i += 2) {
    float ms = 0;
    DSLAM_HIP(hipEventElapsedTime(&ms, e->ev_pool[i], e->ev_pool[i + 1]));
    e->timer_ms += ms;
    e->timer_launches++;
    e->timer_blocks += counts[i / 2];
  }
  e->ev_used = 0;
  if (out_ms) *out_ms = e->timer_ms;
  if (out_launches) *out_launches = e->timer_launches;
  if (out_blocks) *out_blocks = e->timer_blocks;
  return DSLAM_OK;
}

}  // extern "C"
