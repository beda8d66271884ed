// reintegrate_rccl.cpp -- the sharded global re-integration of INTEGRATION.md section 5 as a native program: one process
// per GPU, the C ABI of include/dslam_fusion.h, RCCL (ncclAllGather) for the one exchange (BASELINE configs[4]).
//
//   RANK=r WORLD_SIZE=n DSLAM_NCCL_ID_FILE=/shared/path  reintegrate_rccl frames.bin out.bin [K] [--batched]
//
// Every rank fuses the same N keyframes (frames.bin has driver_harness's layout), then the last K keyframes (default 4)
// get a pose correction (keyframe j of the batch: camera-frame translation (0.01 (j+1), 0, 0.02) m): each is
// de-integrated at its old pose and re-integrated at the new one.  The allocation passes run on every rank (they are
// deterministic and bit-identical, so the hash tables stay equal without communication); voxel blocks are updated only
// by the rank that owns their slot ((slot / 64) % world); one all-gather of the used slot range makes the replicas
// whole again.  With WORLD_SIZE=1 the exchange is a single-rank all-gather: same code path, checked against the
// unsharded result by tests/test_gpu_itmlib_shim.py.  Rank 0 creates the ncclUniqueId and shares it through the file.
// --batched: the keyframes sit in a device-resident keyframe store with the visible lists of their fusion, and the whole
// correction is ONE dslam_reintegrate_batch call per rank (block-major: the allocation passes first, then every touched
// voxel block loaded once); de-integration then visits each keyframe's own stored list.
// Every rank checks the result: the checksums of the hash table and of the voxel blocks are reduced over the ranks with
// ncclAllReduce (min and max); a rank whose replica differs makes every rank exit with status 3.
// out.bin (rank 0): int32 lastFreeBlockId, int32 noVisible, uint64 fnv1a(hash table), uint64 fnv1a(voxel blocks), double ms
// of the re-integration, double ms of the all-gather, uint64 bytes gathered per rank, int32 replicas equal (1 / 0).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dslam_fusion.h"

#define CHECK_DSLAM(x) do { int rc_ = (x); if (rc_ != DSLAM_OK) { fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, dslam_last_error()); return 1; } } while (0)
#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define CHECK_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s failed: %s\n", #x, ncclGetErrorString(r_)); return 1; } } while (0)

static uint64_t fnv1a(const void *p, size_t n, uint64_t h = 1469598103934665603ull) {
  const unsigned char *b = static_cast<const unsigned char *>(p);
  for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}

static int env_int(const char *name, int fallback) {
  const char *v = getenv(name);
  return v ? atoi(v) : fallback;
}

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s frames.bin out.bin [K]\n", argv[0]); return 2; }
  const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), local = env_int("LOCAL_RANK", rank);
  FILE *f = fopen(argv[1], "rb");
  if (!f) { perror("frames"); return 2; }
  int32_t hdr[3];
  if (fread(hdr, 4, 3, f) != 3) return 2;
  const int W = hdr[0], H = hdr[1], N = hdr[2];
  const int K = argc > 3 && argv[3][0] != '-' ? atoi(argv[3]) : (N < 4 ? N : 4);
  bool batched = false;
  for (int a = 3; a < argc; a++) batched = batched || strcmp(argv[a], "--batched") == 0;
  std::vector<std::vector<uint8_t>> rgba(N, std::vector<uint8_t>((size_t)W * H * 4));
  std::vector<std::vector<int16_t>> depth(N, std::vector<int16_t>((size_t)W * H));
  std::vector<std::vector<float>> poses(N, std::vector<float>(16));
  for (int i = 0; i < N; i++)
    if (fread(rgba[i].data(), 1, rgba[i].size(), f) != rgba[i].size() || fread(depth[i].data(), 2, depth[i].size(), f) != depth[i].size() ||
        fread(poses[i].data(), 4, 16, f) != 16) return 2;
  float intr[4], sp[4];
  int32_t ip[4];
  if (fread(intr, 4, 4, f) != 4 || fread(sp, 4, 4, f) != 4 || fread(ip, 4, 4, f) != 4) return 2;
  fclose(f);

  // one rank per GPU
  CHECK_HIP(hipSetDevice(local));
  ncclUniqueId id;
  const char *id_file = getenv("DSLAM_NCCL_ID_FILE");
  if (world > 1 && !id_file) { fprintf(stderr, "WORLD_SIZE > 1 needs DSLAM_NCCL_ID_FILE\n"); return 2; }
  if (rank == 0) {
    CHECK_NCCL(ncclGetUniqueId(&id));
    if (id_file) {
      std::vector<char> tmp(strlen(id_file) + 8);
      snprintf(tmp.data(), tmp.size(), "%s.tmp", id_file);
      FILE *o = fopen(tmp.data(), "wb");
      if (!o || fwrite(&id, sizeof(id), 1, o) != 1) { perror("id file"); return 2; }
      fclose(o);
      rename(tmp.data(), id_file);  // appears atomically for the other ranks
    }
  } else {
    FILE *in = nullptr;
    for (int tries = 0; tries < 600 && !(in = fopen(id_file, "rb")); tries++) usleep(100000);
    if (!in || fread(&id, sizeof(id), 1, in) != 1) { fprintf(stderr, "rank %d: no ncclUniqueId in %s\n", rank, id_file); return 2; }
    fclose(in);
  }
  ncclComm_t comm;
  CHECK_NCCL(ncclCommInitRank(&comm, world, id, rank));

  dslam_engine *eng = nullptr;
  CHECK_DSLAM(dslam_engine_create(local, &eng));
  dslam_scene_params p;
  memset(&p, 0, sizeof(p));
  p.voxel_size = sp[0]; p.mu = sp[1]; p.frustum_min = sp[2]; p.frustum_max = sp[3];
  p.max_w = ip[0]; p.num_local_blocks = ip[1]; p.num_buckets = ip[2]; p.num_excess = ip[3];
  dslam_scene *scene = nullptr;
  dslam_render_state *rs = nullptr;
  dslam_view *view = nullptr;
  CHECK_DSLAM(dslam_scene_create(eng, &p, nullptr, &scene));
  CHECK_DSLAM(dslam_render_state_create(eng, scene, W, H, &rs));
  CHECK_DSLAM(dslam_view_create(eng, W, H, W, H, &view));
  dslam_frame_store *store = nullptr;   // (--batched) mfusionFrameDataBase's images + the visible lists of the fusions
  if (batched) {
    CHECK_DSLAM(dslam_frame_store_create(eng, W, H, W, H, N, &store));
    CHECK_DSLAM(dslam_frame_store_enable_lists(eng, store, scene));
  }
  for (int i = 0; i < N; i++) {  // the live fusion every replica has done
    CHECK_DSLAM(dslam_view_update(eng, view, rgba[i].data(), depth[i].data(), 1e-3f, 0.0f, (double)i, 0));
    if (batched) CHECK_DSLAM(dslam_frame_store_put_view(eng, store, i, view));
    CHECK_DSLAM(dslam_process_frame(eng, scene, view, rs, poses[i].data(), intr, nullptr, nullptr, 0, 0));
    if (batched) CHECK_DSLAM(dslam_frame_store_put_visible_list(eng, store, i, scene, rs));
  }
  if (batched) CHECK_DSLAM(dslam_reintegrate_batch(eng, scene, view, rs, store, 0, nullptr, nullptr, nullptr, intr, 1e-3f, 0.0f));  // (set-up call)

  // ---- the corrected batch, sharded --------------------------------------------------------------------------------
  const int chunk = 64;  // voxel-block slots per ownership chunk (256 KiB)
  const auto t0 = std::chrono::steady_clock::now();
  CHECK_DSLAM(dslam_scene_track_dirty(eng, scene, 1));  // from here on the kernels note every block they visit
  CHECK_DSLAM(dslam_scene_set_shard(scene, rank, world, chunk));
  std::vector<int32_t> slots(K);
  std::vector<float> old_M((size_t)K * 16), new_M((size_t)K * 16);
  for (int j = 0; j < K; j++) {
    const int i = N - K + j;
    std::vector<float> corrected = poses[i];  // column-major world -> camera; a camera-frame translation adds to column 3
    corrected[12] += 0.01f * (float)(j + 1);
    corrected[14] += 0.02f;
    slots[j] = i;
    memcpy(&old_M[(size_t)j * 16], poses[i].data(), 64);
    memcpy(&new_M[(size_t)j * 16], corrected.data(), 64);
    if (batched) continue;
    CHECK_DSLAM(dslam_view_update(eng, view, rgba[i].data(), depth[i].data(), 1e-3f, 0.0f, (double)i, 0));
    CHECK_DSLAM(dslam_deprocess_frame(eng, scene, view, rs, poses[i].data(), intr, nullptr, nullptr));
    CHECK_DSLAM(dslam_process_frame(eng, scene, view, rs, corrected.data(), intr, nullptr, nullptr, 0, /*isDefusion*/ 1));
  }
  if (batched) CHECK_DSLAM(dslam_reintegrate_batch(eng, scene, view, rs, store, K, slots.data(), old_M.data(), new_M.data(), intr, 1e-3f, 0.0f));
  dslam_stats st;
  CHECK_DSLAM(dslam_get_stats(eng, scene, rs, &st));  // synchronises; identical on every rank
  const auto t1 = std::chrono::steady_clock::now();
  // exactly the blocks the batch touched: every rank derives the same per-shard lists, so neither ids nor counts travel
  std::vector<int32_t> counts(world);
  CHECK_DSLAM(dslam_shard_dirty_plan(eng, scene, world, chunk, counts.data()));
  int cap = 1;
  for (int c : counts) cap = c > cap ? c : cap;
  size_t gathered = 0;
  {
    const size_t bytes_per_rank = (size_t)cap * 4096;
    void *send = nullptr, *recv = nullptr;
    CHECK_HIP(hipMalloc(&send, bytes_per_rank));
    CHECK_HIP(hipMalloc(&recv, bytes_per_rank * world));
    hipStream_t stream = static_cast<hipStream_t>(dslam_engine_stream(eng));
    CHECK_DSLAM(dslam_shard_dirty_pack(eng, scene, rank, send, cap));
    CHECK_NCCL(ncclAllGather(send, recv, bytes_per_rank, ncclChar, comm, stream));  // (same stream: ordered behind the pack)
    CHECK_DSLAM(dslam_shard_dirty_unpack(eng, scene, rank, recv, cap));
    CHECK_DSLAM(dslam_engine_synchronize(eng));
    CHECK_HIP(hipFree(send));
    CHECK_HIP(hipFree(recv));
    gathered = bytes_per_rank;
  }
  CHECK_DSLAM(dslam_scene_track_dirty(eng, scene, 0));
  CHECK_DSLAM(dslam_scene_set_shard(scene, 0, 1, chunk));  // back to single-GPU fusion
  const auto t2 = std::chrono::steady_clock::now();

  std::vector<dslam_hash_entry> hash((size_t)p.num_buckets + p.num_excess);
  std::vector<dslam_voxel> vox((size_t)p.num_local_blocks * 512);
  CHECK_DSLAM(dslam_download_hash_table(eng, scene, hash.data()));
  CHECK_DSLAM(dslam_download_voxel_blocks(eng, scene, 0, p.num_local_blocks, vox.data()));
  const double ms_reint = std::chrono::duration<double, std::milli>(t1 - t0).count();
  const double ms_gather = std::chrono::duration<double, std::milli>(t2 - t1).count();
  // every rank's replica must be the same map: min and max of the checksums over the ranks
  int32_t replicas_equal = 1;
  {
    const uint64_t mine[2] = {fnv1a(hash.data(), hash.size() * sizeof(dslam_hash_entry)), fnv1a(vox.data(), vox.size() * sizeof(dslam_voxel))};
    uint64_t *dev = nullptr, lo[2], hi[2];
    CHECK_HIP(hipMalloc(&dev, 4 * sizeof(uint64_t)));
    CHECK_HIP(hipMemcpy(dev, mine, sizeof(mine), hipMemcpyHostToDevice));
    hipStream_t stream = static_cast<hipStream_t>(dslam_engine_stream(eng));
    CHECK_NCCL(ncclAllReduce(dev, dev + 2, 2, ncclUint64, ncclMin, comm, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_HIP(hipMemcpy(lo, dev + 2, sizeof(lo), hipMemcpyDeviceToHost));
    CHECK_NCCL(ncclAllReduce(dev, dev + 2, 2, ncclUint64, ncclMax, comm, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_HIP(hipMemcpy(hi, dev + 2, sizeof(hi), hipMemcpyDeviceToHost));
    CHECK_HIP(hipFree(dev));
    if (lo[0] != hi[0] || lo[1] != hi[1]) {
      replicas_equal = 0;
      fprintf(stderr, "rank %d: replicas differ after the exchange (hash table %s, voxel blocks %s; mine %016llx %016llx)\n", rank,
              lo[0] == hi[0] ? "equal" : "DIFFERENT", lo[1] == hi[1] ? "equal" : "DIFFERENT", (unsigned long long)mine[0], (unsigned long long)mine[1]);
    }
  }
  if (rank == 0) {
    FILE *o = fopen(argv[2], "wb");
    if (!o) { perror("out"); return 2; }
    const int32_t head[2] = {st.last_free_block_id, st.no_visible_entries};
    const uint64_t sums[2] = {fnv1a(hash.data(), hash.size() * sizeof(dslam_hash_entry)), fnv1a(vox.data(), vox.size() * sizeof(dslam_voxel))};
    const uint64_t g = gathered;
    fwrite(head, 4, 2, o); fwrite(sums, 8, 2, o); fwrite(&ms_reint, 8, 1, o); fwrite(&ms_gather, 8, 1, o); fwrite(&g, 8, 1, o);
    fwrite(&replicas_equal, 4, 1, o);
    fclose(o);
    printf("reintegrate_rccl %s: world %d, %d keyframes re-integrated%s in %.3f ms, all-gather of %zu bytes per rank in %.3f ms\n",
           replicas_equal ? "ok" : "FAILED (replicas differ)", world, K, batched ? " (one dslam_reintegrate_batch call)" : "", ms_reint, gathered, ms_gather);
  }
  if (store) dslam_frame_store_destroy(store);
  dslam_view_destroy(view);
  dslam_render_state_destroy(rs);
  dslam_scene_destroy(scene);
  dslam_engine_destroy(eng);
  ncclCommDestroy(comm);
  return replicas_equal ? 0 : 3;
}
