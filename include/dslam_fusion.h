/*
 * dslam_fusion.h -- C ABI of libdslam_fusion.so, the MI355X (gfx950) TSDF fusion + raycast engine.
 *
 * This is the drop-in boundary for the voxel-block-hashing hot path of
 * Hansry/DenseSLAM-Global-Consistency-h.  Every entry point names the ITMLib engine method it stands in
 * for and the reference call site that reaches it (file:line under /root/reference/src/DenseSLAM).  The
 * ITMLib-compatible C++ classes in denseslam-global-consistency-h_amd/itmlib/ (ITMDenseMapper,
 * ITMSceneReconstructionEngine, ITMVisualisationEngine, ITMSwappingEngine, ...) are thin wrappers over
 * these functions, so InfiniTamDriver / DenseSlam / DenseSLAMGUI compile and run unchanged on top.
 *
 * Conventions
 *  - plain C, no exceptions across the boundary; every function returns a dslam_status (0 = ok).
 *  - 4x4 matrices are float[16], COLUMN-major, exactly ORUtils::Matrix4f::m[] (InfiniTamDriver.cpp:208-226).
 *  - intrinsics are float[4] = (fx, fy, cx, cy) = ITMIntrinsics::projectionParamsSimple.all
 *    (InfiniTamDriver.cpp:60-67).
 *  - "host" pointers are ordinary CPU memory, "dev" pointers are HIP device memory on the engine's device.
 *  - the engine is used from one thread (DenseSlam's main thread); it owns one HIP stream.
 *    In the default synchronous mode every call has completed (including D2H copies) when it returns,
 *    which is what the reference callers assume (DenseSlam.h:151-152,162-163).  dslam_engine_set_async
 *    lets a caller pipeline frames and synchronise explicitly (the ITMLib mirror in itmlib/ runs the engine this way
 *    and synchronises where data is handed to the host).
 *  - conditions only the device can detect (an allocation ray longer than the order key encodes: DSLAM_ERR_UNSUPPORTED;
 *    a tile count of an ordered compaction that never arrived, after which the map state is undefined: DSLAM_ERR_HIP)
 *    are returned by the first call that waits for the stream after the kernel in question: the call itself on a
 *    synchronous engine; on an asynchronous one the next dslam_engine_synchronize, dslam_fence_wait, dslam_get_stats,
 *    read-back (dslam_download_*, an image into ordinary host memory, dslam_mesh_download ...).  Told once per
 *    occurrence; dslam_get_stats keeps reporting them for the scene they happened in until dslam_scene_reset.
 */
#ifndef DSLAM_FUSION_H
#define DSLAM_FUSION_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSLAM_BLOCK_SIZE 8    /* SDF_BLOCK_SIZE  */
#define DSLAM_BLOCK_SIZE3 512 /* SDF_BLOCK_SIZE3, used at InfiniTamDriver.h:346 */

/* upstream ITMLibDefines.h defaults (corroborated by the reference's memory logs) */
#define DSLAM_DEFAULT_LOCAL_BLOCK_NUM 0x40000 /* SDF_LOCAL_BLOCK_NUM   */
#define DSLAM_DEFAULT_BUCKET_NUM 0x100000     /* SDF_BUCKET_NUM        */
#define DSLAM_DEFAULT_EXCESS_LIST_SIZE 0x20000 /* SDF_EXCESS_LIST_SIZE */
#define DSLAM_TRANSFER_BLOCK_NUM 0x1000       /* SDF_TRANSFER_BLOCK_NUM */
#define DSLAM_MAX_RENDERING_BLOCKS (65536 * 4)

typedef enum {
  DSLAM_OK = 0,
  DSLAM_ERR_INVALID = -1,     /* bad argument (null handle, size mismatch, ...) */
  DSLAM_ERR_HIP = -2,         /* a HIP runtime call failed; see dslam_last_error() */
  DSLAM_ERR_UNSUPPORTED = -3, /* parameter combination outside what the kernels are built for */
  DSLAM_ERR_NO_DEVICE = -4
} dslam_status;

/* ITMHashEntry {Vector3s pos; int offset; int ptr;}  -- 16 bytes.
 * ptr >= 0: slot in the voxel block array; ptr == -1: swapped out; ptr < -1: unused entry.
 * offset >= 1: next entry of the bucket lives at num_buckets + offset - 1. */
typedef struct {
  int16_t pos[3];
  int16_t _pad;
  int32_t offset;
  int32_t ptr;
} dslam_hash_entry;

/* ITMVoxel = ITMVoxel_s_rgb {short sdf; uchar w_depth; Vector3u clr; uchar w_color;} -- 8 bytes
 * (sizeof(ITMVoxel) at InfiniTamDriver.h:333-335).  Empty voxel: sdf 32767, everything else 0. */
typedef struct {
  int16_t sdf;
  uint8_t w_depth;
  uint8_t clr[3];
  uint8_t w_color;
  uint8_t _pad;
} dslam_voxel;

/* ITMSceneParams + the compile-time pool sizes of ITMLibDefines.h made runtime fields. */
typedef struct {
  float voxel_size;   /* metres                     (upstream default 0.005) */
  float mu;           /* truncation band, metres    (0.02) */
  int32_t max_w;      /* weight clamp, <= 255       (100)  */
  float frustum_min;  /* viewFrustum_min, metres    (0.2)  */
  float frustum_max;  /* viewFrustum_max, metres    (3.0)  */
  int32_t stop_integrating_at_max_w;
  int32_t num_local_blocks; /* SDF_LOCAL_BLOCK_NUM; 0 -> default */
  int32_t num_buckets;      /* SDF_BUCKET_NUM, power of two; 0 -> default */
  int32_t num_excess;       /* SDF_EXCESS_LIST_SIZE; 0 -> default */
  int32_t use_swapping;     /* ITMLibSettings::useSwapping: allocate the host-side global cache */
  int32_t history_words;    /* 64-bit words per visible-list ring per block (ring holds 64*words lists,
                               bounds max_age / defusion maxSize); 0 -> 4 */
} dslam_scene_params;

/* ITMLib::Engine::WeightParams {depthWeighting, maxNewW, maxDistance} (SystemEntry.cpp:183-187,
 * InfiniTamDriver.h:189,196). */
typedef struct {
  int32_t depth_weighting;
  int32_t max_new_w;   /* 1..255 (a voxel weight is one byte; the kernels tabulate 1/(w_depth + newW)) */
  float max_distance;
} dslam_weight_params;

/* What InfiniTamDriver reads back after every call (InfiniTamDriver.h:209-210,344-351,366-370). */
typedef struct {
  int32_t num_allocated_blocks; /* scene->index.getNumAllocatedVoxelBlocks() == num_local_blocks */
  int32_t last_free_block_id;   /* scene->localVBA.lastFreeBlockId */
  int32_t last_free_excess_id;  /* scene->index.lastFreeExcessListId */
  int32_t no_visible_entries;   /* ITMRenderState_VH::noVisibleEntries of the render state passed */
  int64_t decayed_block_count;  /* ITMDenseMapper::GetDecayedBlockCount() */
  int64_t slid_block_count;     /* blocks released (or swapped out) by SlideWindow* so far */
  int32_t frame_counter;        /* number of visible lists queued so far (fusion + defusion) */
  int32_t fusion_fifo_len;
  int32_t defusion_fifo_len;
  int32_t alloc_failures;       /* blocks the last allocation wanted but could not get (pool exhausted) */
  int32_t last_swapped_in;      /* blocks moved by the last IntegrateGlobalIntoLocal */
  int32_t last_swapped_out;     /* blocks moved by the last SaveToGlobalMemory */
} dslam_stats;

/* ITMMainEngine::GetImageType values used by the reference (InfiniTamDriver.cpp:16-38). */
typedef enum {
  DSLAM_IMAGE_SHADED = 0,             /* InfiniTAM_IMAGE_FREECAMERA_SHADED */
  DSLAM_IMAGE_COLOUR_FROM_VOLUME = 1, /* InfiniTAM_IMAGE_FREECAMERA_COLOUR_FROM_VOLUME */
  DSLAM_IMAGE_COLOUR_FROM_NORMAL = 2, /* InfiniTAM_IMAGE_FREECAMERA_COLOUR_FROM_NORMAL */
  DSLAM_IMAGE_DEPTH = 3               /* InfiniTAM_IMAGE_FREECAMERA_DEPTH (float metres) */
} dslam_image_type;

typedef struct dslam_engine dslam_engine;             /* device + stream + scratch: the *_HIP engines */
typedef struct dslam_scene dslam_scene;               /* ITMScene<ITMVoxel,ITMVoxelBlockHash> (+ITMGlobalCache) */
typedef struct dslam_render_state dslam_render_state; /* ITMRenderState_VH */
typedef struct dslam_view dslam_view;                 /* ITMView (rgb, depth) */
typedef struct dslam_fence dslam_fence;               /* a marker in the engine's stream (async mode) */

/* ---- engine ------------------------------------------------------------------------------------ */
const char *dslam_last_error(void);
const char *dslam_version(void);
/* ITMSceneReconstructionEngineFactory / ITMVisualisationEngineFactory / ITMSwappingEngineFactory for
 * the HIP device type (ITMMainEngine ctor, InfiniTamDriver.h:102). */
int dslam_engine_create(int device_index, dslam_engine **out);
/* The NUMA node of the host the device hangs off (-1 if the platform does not say; read from sysfs by PCI bus id).  A caller
 * that fills page-locked images every frame (CvToItm, InfiniTamDriver.cpp:17-75) should run on that node: the 1.8 MB fill of a
 * 640x480 frame takes ~60 us there and ~160 us from the other socket of a two-socket host.  May be
 * called before any engine exists. */
int dslam_device_numa_node(int device_index, int *node_out);
int dslam_engine_destroy(dslam_engine *e);
int dslam_engine_set_async(dslam_engine *e, int async_mode);
/* waits for everything enqueued so far; returns what kernels reported since the last synchronising call (above) */
int dslam_engine_synchronize(dslam_engine *e);
/* native hipStream_t of the engine, for callers that enqueue their own work (RCCL, torch). */
void *dslam_engine_stream(dslam_engine *e);
/* Pipelining across PCIe (async mode only; the reference's own driver is synchronous and needs none of this).
 * With dslam_engine_set_async(e, 1) and caller images in dslam_host_alloc memory:
 *  - dslam_view_update copies on a second (copy) stream into one of two landing buffers of the view while the kernels
 *    of the previous frame, enqueued earlier, keep the GPU busy; it returns when the frame has landed (the calling
 *    thread sits out the copy, the GPU does not).  A frame whose depth image directly follows its RGBA image in
 *    memory goes up as one copy.
 *  - dslam_get_image into a page-locked image returns at once; the render kernel stores the pixels there itself.
 *  - a fence marks "everything enqueued on the engine so far": record it after a frame's last call, wait for it
 *    before reading that frame's output image or rewriting its input images. */
int dslam_fence_create(dslam_engine *e, dslam_fence **out);
int dslam_fence_destroy(dslam_fence *f);
int dslam_fence_record(dslam_engine *e, dslam_fence *f);
int dslam_fence_wait(dslam_fence *f);               /* blocks the calling thread; a fence never recorded has passed */
int dslam_fence_query(dslam_fence *f, int *done);   /* non-blocking */
/* Page-locked host memory for the caller's image buffers: what ORUtils::MemoryBlock's host side is whenever the
 * block also has a device side (upstream ORUtils/MemoryBlock.h Allocate: cudaMallocHost), i.e. the rgb / raw-depth
 * images DenseSlam::ProcessFrame fills each frame (DenseSlam.cpp:66-74).  Zero-filled.  In the default synchronous
 * mode dslam_view_update* DMA straight from such buffers; any other host pointer is staged through the engine's
 * own pinned buffer first.  Needs no engine (the images are created before it). */
int dslam_host_alloc(size_t bytes, void **out);
int dslam_host_free(void *p);
/* Self-test: the integration kernel divides with a 2-wide, scaling-free form of the hardware's IEEE division sequence
 * (csrc/integrate.hip div_ieee2).  Compares it with the native float division on `samples` random operand pairs
 * drawn from the kernel's operand ranges, on the device; *mismatches_out must come back 0. */
int dslam_selftest_division(dslam_engine *e, long long samples, long long *mismatches_out);
/* Test hook: CreateExpectedDepths' render-tile budget (MAX_RENDERING_BLOCKS, default DSLAM_MAX_RENDERING_BLOCKS).
 * Upstream drops, in visible-list order, every block whose tiles would reach the budget; real scenes never get
 * there (it takes > 262144 tiles), so the parity test of that rule lowers the budget instead. */
int dslam_debug_set_render_tile_budget(dslam_engine *e, int budget);
/* Test hook: ProcessFrame queues the frame's visible list on the ring either from the fusion kernel's block waves or -- from
 * this many visible blocks on (default 65536: maps whose visible voxels no longer fit the Infinity Cache) -- from extra
 * workgroups at the end of the same launch (csrc/integrate.hip, kPushJobMin).  From the same size on the fusion and
 * de-integration launches over a render state's list read and write their voxel blocks with the non-temporal cache policy
 * (chosen by the host from the visible count the last allocation pass reported).  Same bits either way; the parity test of
 * the second forms lowers the threshold instead of building a quarter-million-block scene for the oracle. */
int dslam_debug_set_push_job_min(dslam_engine *e, int min_visible_blocks);
/* Test hook: how many fusion / de-integration launches of this engine took the streaming (non-temporal) instantiation. */
int dslam_debug_stream_launches(dslam_engine *e, long long *count_out);
/* Test hook: a one-thread kernel reports the given device-side error bits for the scene (1: allocation ray longer than the
 * order key encodes, 2: a tile count never arrived) exactly as a failing pass would (report_error, csrc/dslam_device.h), so
 * that the way such an error reaches the caller can be tested: returned by this very call on a synchronous engine, by the
 * next call that waits for the stream on an asynchronous one -- once --, and by dslam_get_stats of that scene until it is
 * reset. */
int dslam_debug_inject_device_error(dslam_engine *e, dslam_scene *s, int bits);

/* ---- scene ------------------------------------------------------------------------------------- */
/* new ITMScene(sceneParams, useSwapping, memoryType) + ResetScene.  ext_voxel_blocks_dev may be NULL
 * (library allocates) or a caller-owned device buffer of num_local_blocks*512*8 bytes (e.g. a torch
 * tensor used as the RCCL all-gather buffer). */
int dslam_scene_create(dslam_engine *e, const dslam_scene_params *p, void *ext_voxel_blocks_dev,
                       dslam_scene **out);
int dslam_scene_destroy(dslam_scene *s);
/* denseMapper->ResetScene(scene)  (InfiniTamDriver.h:354-360). */
int dslam_scene_reset(dslam_engine *e, dslam_scene *s);
int dslam_scene_get_params(const dslam_scene *s, dslam_scene_params *out);

/* ---- render state / view ----------------------------------------------------------------------- */
/* visualisationEngine->CreateRenderState(imgSize) */
int dslam_render_state_create(dslam_engine *e, const dslam_scene *s, int width, int height,
                              dslam_render_state **out);
int dslam_render_state_destroy(dslam_render_state *r);
int dslam_view_create(dslam_engine *e, int width_rgb, int height_rgb, int width_d, int height_d,
                      dslam_view **out);
int dslam_view_destroy(dslam_view *v);

/* viewBuilder->UpdateView(&view, rgb, rawDepth, timestamp, useBilateralFilter)
 * (InfiniTamDriver.cpp:280-288).  rgba: Vector4u per pixel as written by CvToItm (:84-103);
 * depth_mm: int16 millimetres (:106-110); depth_m = d<=0 ? -1 : d*affine_a + affine_b with
 * (a,b) = (1/1000, 0) from CreateItmCalib (:58,79).
 * use_bilateral_filter: ITMViewBuilder's five passes of the 5x5 bilateral depth filter (upstream InfiniTAM v2
 * filterDepth); the filtered image keeps upstream's 2-pixel border of 0 (= no measurement). */
int dslam_view_update(dslam_engine *e, dslam_view *v, const uint8_t *rgba_host, const int16_t *depth_mm_host,
                      float affine_a, float affine_b, double timestamp, int use_bilateral_filter);
/* same, inputs already resident in HBM (frame database kept on device). */
int dslam_view_update_device(dslam_engine *e, dslam_view *v, const void *rgba_dev, const void *depth_mm_dev,
                             float affine_a, float affine_b, double timestamp, int use_bilateral_filter);

/* CvToItm(const cv::Mat3b&, ITMUChar4Image*) fused into UpdateView (InfiniTamDriver.cpp:84-103, 280-288): the
 * colour image arrives as OpenCV packed BGR (3 bytes per pixel, rows contiguous) and is converted to RGBA with
 * a = 255 on the device.  The _device variant needs a 4-byte aligned image. */
int dslam_view_update_bgr(dslam_engine *e, dslam_view *v, const uint8_t *bgr_host, const int16_t *depth_mm_host,
                          float affine_a, float affine_b, double timestamp, int use_bilateral_filter);
int dslam_view_update_bgr_device(dslam_engine *e, dslam_view *v, const void *bgr_dev, const void *depth_mm_dev,
                                 float affine_a, float affine_b, double timestamp, int use_bilateral_filter);
/* Dataset wire formats either side of the path, converted on the device.
 * Input: the per-pixel loop of PrecomputedDepthProvider::ReadPrecomputed on the 16-bit depth image as stored by the
 * datasets (PrecomputedDepthProvider.cpp:30-64): KITTI-style maps hold depth * 256 (values above max_depth_m * 256
 * are dropped, then int16 mm = (int16)((float)v * (1000 / 256))), TUM / ICL-NUIM maps are divided by 5.0 and
 * dropped above (int16)round(max_depth_m * 1000).  The converted millimetre image then takes UpdateView's path.
 * colour_channels: 4 = RGBA as dslam_view_update, 3 = OpenCV BGR as dslam_view_update_bgr.  (A float -> int16
 * conversion that overflows is undefined in C; here it wraps like the x86 code the reference compiles to.) */
enum { DSLAM_DEPTH_MM = 0, DSLAM_DEPTH_KITTI_X256 = 1, DSLAM_DEPTH_RGBD_X5 = 2 };
int dslam_view_update_dataset(dslam_engine *e, dslam_view *v, const uint8_t *colour_host, int colour_channels,
                              const int16_t *depth_raw_host, int depth_format, float max_depth_m, float affine_a,
                              float affine_b, double timestamp, int use_bilateral_filter);
/* read-back of the view's raw millimetre image as the kernels see it (after the dataset conversion) */
int dslam_download_view_raw_depth(dslam_engine *e, const dslam_view *v, int16_t *out_mm);
/* Output: GetImage(FREECAMERA_DEPTH) followed by FloatDepthmapToShort (scale 1000, InfiniTamDriver.cpp:167-180) or
 * FloatDepthmapToInt16 (scale 256, the raycast-depth PNGs of DenseSlam::SaveRaycastDepth, InfiniTamDriver.cpp:188-
 * 200, DenseSlam.cpp:573-592): int16 = (int16)(depth_m * scale) per pixel, converted on the device, so half the
 * bytes cross PCIe and the host loop disappears. */
int dslam_get_depth_image_int16(dslam_engine *e, const dslam_scene *s, dslam_render_state *r, const float M[16],
                                const float intrinsics[4], int scale, int16_t *out_host);
/* test / debug read-back of the view's RGBA image (what IntegrateIntoScene will read). */
int dslam_download_view_rgba(dslam_engine *e, const dslam_view *v, uint8_t *out_rgba);

/* ---- keyframe store --------------------------------------------------------------------------- */
/* The image payload of DenseSlam's mfusionFrameDataBase (fusionFrameInfo::rgbinfo / depthinfo, DenseSlam.h:431-433)
 * kept resident in HBM: `capacity` slots of one RGBA + one int16-millimetre depth image.  The
 * reference re-uploads every keyframe through UpdateView for each de-/re-integration (DenseSlam.cpp:389-403,
 * 420-422); with the store OnlineCorrection's UpdateView becomes dslam_view_update_from_store -- no copy at all.
 * Slot bookkeeping (timestamp -> slot) stays with the caller's std::map.  640x480: 1.8 MB per keyframe. */
typedef struct dslam_frame_store dslam_frame_store;
int dslam_frame_store_create(dslam_engine *e, int width_rgb, int height_rgb, int width_d, int height_d, int capacity,
                             dslam_frame_store **out);
int dslam_frame_store_destroy(dslam_frame_store *fs);
/* host images in (the same layouts as dslam_view_update / dslam_view_update_bgr) */
int dslam_frame_store_put(dslam_engine *e, dslam_frame_store *fs, int slot, const uint8_t *rgba_host,
                          const int16_t *depth_mm_host);
int dslam_frame_store_put_bgr(dslam_engine *e, dslam_frame_store *fs, int slot, const uint8_t *bgr_host,
                              const int16_t *depth_mm_host);
/* device-to-device from the view that was just fused (mfusionFrameDataBase insert, DenseSlam.cpp:196-208):
 * the keyframe never crosses PCIe a second time */
int dslam_frame_store_put_view(dslam_engine *e, dslam_frame_store *fs, int slot, const dslam_view *v);
int dslam_frame_store_get(dslam_engine *e, const dslam_frame_store *fs, int slot, uint8_t *rgba_out,
                          int16_t *depth_mm_out);
int dslam_frame_store_device_ptrs(const dslam_frame_store *fs, int slot, void **rgba_dev, void **depth_mm_dev);
/* static_scene_->UpdateView(currRGBInfo, currDepthInfo, timestamp) for a stored keyframe (DenseSlam.cpp:392,421):
 * the view reads the slot in place until its next update; overwriting the slot before that changes what it sees */
int dslam_view_update_from_store(dslam_engine *e, dslam_view *v, const dslam_frame_store *fs, int slot,
                                 float affine_a, float affine_b, double timestamp, int use_bilateral_filter);

/* The blocks a keyframe was fused into, kept with it (optional).  DeProcessFrame as the reference calls it has only the
 * frame and its old pose, so it first has to find the blocks again: a visible-list-only allocation pass at that pose (two
 * kernel launches over the depth image and the table) -- which in a re-integration batch is work every GPU repeats.  With
 * the list stored at fusion time, de-integration goes straight to the integration kernel:
 *   dslam_frame_store_enable_lists(e, fs, scene)            room for one list of num_local_blocks entries per slot; the
 *                                                           lists hold entry ids of THIS scene's table size, and every
 *                                                           call that reads one refuses a scene of another size
 *   dslam_frame_store_put_visible_list(e, fs, slot, s, r)   after ProcessFrame of that keyframe (fusion or re-fusion): the
 *                                                           render state's visible list with each entry's block position
 *   dslam_deprocess_frame_stored(e, s, v, fs, slot, M_d, ...)   the inverse update of DeProcessFrame on exactly those
 *                                                           blocks: an entry that no longer holds the same block (released
 *                                                           by decay / the window since, or re-used) is skipped
 * Semantics differ from dslam_deprocess_frame in WHICH blocks are visited (the keyframe's own, instead of whatever an
 * allocation pass at the old pose finds today, previous-list carry-over included), and the render state is left alone.
 * The view must hold the keyframe's images (dslam_view_update_from_store). */
int dslam_frame_store_enable_lists(dslam_engine *e, dslam_frame_store *fs, const dslam_scene *s);
int dslam_frame_store_put_visible_list(dslam_engine *e, dslam_frame_store *fs, int slot, const dslam_scene *s,
                                       const dslam_render_state *r);
int dslam_deprocess_frame_stored(dslam_engine *e, dslam_scene *s, const dslam_view *v, const dslam_frame_store *fs, int slot,
                                 const float M_d[16], const float intrinsics_d[4], const float M_rgb[16],
                                 const float intrinsics_rgb[4]);

/* The re-integration batch of DenseSlam::OnlineCorrection (DenseSlam.cpp:389-403: for every corrected keyframe
 * DeProcessFrame at its old pose, ProcessFrame at the new one), as ONE call.  Equal -- map, rings, free lists, render state,
 * stored lists: bit for bit -- to
 *   for k in 0 .. n-1:  dslam_view_update_from_store(v, fs, slots[k]);  dslam_deprocess_frame_stored(s, v, fs, slots[k], old_M[k]);
 *                       dslam_process_frame(s, v, r, new_M[k], is_defusion = 1);  dslam_frame_store_put_visible_list(fs, slots[k], s, r)
 * but run block-major: the n allocation passes first, then every voxel block the batch touches is loaded once, takes the
 * de- and re-updates of every keyframe that names it in keyframe order, and is stored once (csrc/integrate.hip).  The
 * keyframes' images and fusion-time visible lists must be in the store; poses are n x 16 floats (column-major world ->
 * camera), one camera (depth = colour).  Scenes with host swapping or stopIntegratingAtMaxW return DSLAM_ERR_UNSUPPORTED:
 * use the per-keyframe calls.  On a sharded scene (dslam_scene_set_shard) every rank runs the allocation passes and
 * updates its own blocks; the exchange is dslam_shard_dirty_plan / _pack / _unpack as for the per-keyframe calls.
 * n = 0 changes nothing and allocates the call's scratch buffers (a second copy of 32 stored lists, per-block operation
 * masks): a set-up call keeps those allocations out of the first batch.
 * Errors: every argument -- slots, stored lists, table sizes, invertible new poses, the order key's range -- is checked
 * before anything is changed, so DSLAM_ERR_INVALID / DSLAM_ERR_UNSUPPORTED leave scene, render state and store as they
 * were.  A HIP failure in mid-batch (DSLAM_ERR_HIP) does not: allocation passes may have run whose blocks were never de-
 * or re-integrated, i.e. the scene is UNDEFINED (neither the old nor the new map) and must be reset or restored; the
 * render state's visible list is re-derived by its next pass, and the keyframes of the interrupted chunk lose their
 * stored lists (dslam_frame_store_put_visible_list again after re-fusing them). */
int dslam_reintegrate_batch(dslam_engine *e, dslam_scene *s, dslam_view *v, dslam_render_state *r, dslam_frame_store *fs,
                            int n, const int32_t *slots, const float *old_M, const float *new_M, const float intr[4],
                            float affine_a, float affine_b);

/* what the last dslam_reintegrate_batch (its last chunk of <= 32 keyframes) worked on: the distinct voxel blocks it loaded, and
 * its block-operations -- (block, keyframe) pairs de-integrated or re-fused, the unit the block kernel's cost is quoted in */
int dslam_reintegrate_batch_stats(dslam_engine *e, const dslam_scene *s, int32_t *blocks_out, int32_t *block_operations_out);

/* DenseSlam::depthPostProcessing's pixel loop (DenseSlam.cpp:488-529): blanks (sets to 0) every pixel of the
 * current keyframe's depth whose reprojection into the previous keyframe disagrees with that keyframe's depth by
 * more than `filter_threshold` (relative) and which lies below row `filter_area * rows`
 * (PostPocessParams, VoxelDecayParams.h:38-46).  Tpc = prev_pose.inv() * curr_pose (DenseSlam.cpp:507), column-major
 * like every matrix of this ABI; intrinsics = (fx, fy, cx, cy) of projection_left_rgb_ (:436-439).  The reference
 * pairs `row` with cx/fx and `col` with cy/fy (:500-513); that pairing is reproduced.  curr is updated in place;
 * *count_out (optional) receives the reference's `count` (pixels compared).  Host variant: cv::Mat1s buffers;
 * device variant: frames resident in HBM (frame database on device). */
int dslam_depth_post_processing(dslam_engine *e, int16_t *curr_depth_mm_host, const int16_t *prev_depth_mm_host,
                                int width, int height, const float Tpc[16], const float intrinsics[4],
                                float filter_threshold, float filter_area, int *count_out);
int dslam_depth_post_processing_device(dslam_engine *e, void *curr_depth_mm_dev, const void *prev_depth_mm_dev,
                                       int width, int height, const float Tpc[16], const float intrinsics[4],
                                       float filter_threshold, float filter_area, int *count_out);

/* ---- fusion ------------------------------------------------------------------------------------ */
/* denseMapper->SetFusionWeightParams(...)  (InfiniTamDriver.h:189,196) */
int dslam_set_fusion_weight_params(dslam_engine *e, const dslam_weight_params *w);

/* sceneRecoEngine->AllocateSceneFromDepth(scene, view, trackingState, renderState, onlyUpdateVisibleList)
 * M_d = trackingState->pose_d->GetM() (world -> camera), InfiniTamDriver.h:173-178. */
int dslam_allocate_scene_from_depth(dslam_engine *e, dslam_scene *s, const dslam_view *v,
                                    dslam_render_state *r, const float M_d[16], const float intrinsics_d[4],
                                    int only_update_visible_list);
/* sceneRecoEngine->IntegrateIntoScene(scene, view, trackingState, renderState).
 * M_rgb = calib.trafo_rgb_to_depth.calib_inv * M_d (identity calib in the reference,
 * InfiniTamDriver.cpp:74-75); pass M_rgb = NULL for "same as M_d". */
int dslam_integrate_into_scene(dslam_engine *e, dslam_scene *s, const dslam_view *v,
                               const dslam_render_state *r, const float M_d[16],
                               const float intrinsics_d[4], const float M_rgb[16],
                               const float intrinsics_rgb[4]);
/* denseMapper->ProcessFrame(view, trackingState, scene, renderState, onlyUpdateVisibleList, isDefusion)
 * (InfiniTamDriver.h:187-192; DenseSlam.cpp:213,236,403): allocate + integrate, queue the frame's
 * visible list (fusion or defusion FIFO), then swap in/out when the scene was created with swapping. */
int dslam_process_frame(dslam_engine *e, dslam_scene *s, const dslam_view *v, dslam_render_state *r,
                        const float M_d[16], const float intrinsics_d[4], const float M_rgb[16],
                        const float intrinsics_rgb[4], int only_update_visible_list, int is_defusion);
/* denseMapper->DeProcessFrame(view, trackingState, scene, renderState)
 * (InfiniTamDriver.h:194-199; DenseSlam.cpp:393,425): visible-list-only pass at the old pose, then the
 * inverse running average. */
int dslam_deprocess_frame(dslam_engine *e, dslam_scene *s, const dslam_view *v, dslam_render_state *r,
                          const float M_d[16], const float intrinsics_d[4], const float M_rgb[16],
                          const float intrinsics_rgb[4]);

/* ---- map regularisation / sliding window (the "global consistency" memory path) ------------------ */
/* denseMapper->Decay(scene, renderState, maxWeight, minAge, forceAllVoxels) (InfiniTamDriver.h:274-282,
 * 315-331) and DecayDefusionPart (:284-292). */
int dslam_decay(dslam_engine *e, dslam_scene *s, dslam_render_state *r, int max_weight, int min_age,
                int force_all_voxels);
int dslam_decay_defusion_part(dslam_engine *e, dslam_scene *s, dslam_render_state *r, int max_weight,
                              int min_age, int force_all_voxels);
/* denseMapper->SlideWindow(scene, renderState, maxAge) (InfiniTamDriver.h:294-300) and
 * SlideWindowDefusionPart(scene, renderState, maxAge, maxSize) (:302-310). */
int dslam_slide_window(dslam_engine *e, dslam_scene *s, dslam_render_state *r, int max_age);
int dslam_slide_window_defusion_part(dslam_engine *e, dslam_scene *s, dslam_render_state *r, int max_age,
                                     int max_size);

/* ---- swapping ---------------------------------------------------------------------------------- */
/* ITMSwappingEngine::IntegrateGlobalIntoLocal(scene, renderState) */
int dslam_swap_in(dslam_engine *e, dslam_scene *s, dslam_render_state *r);
/* ITMSwappingEngine::SaveToGlobalMemory(scene, renderState): swap out blocks that left the view */
int dslam_swap_out(dslam_engine *e, dslam_scene *s, dslam_render_state *r);
/* Hansry's one-argument SaveToGlobalMemory(scene) (DenseSlam.h:248-251): flush every resident block */
int dslam_save_to_global_memory(dslam_engine *e, dslam_scene *s);

/* ---- visualisation / raycast ------------------------------------------------------------------- */
/* visualisationEngine->FindVisibleBlocks(scene, pose, intrinsics, renderState) */
int dslam_find_visible_blocks(dslam_engine *e, const dslam_scene *s, dslam_render_state *r,
                              const float M[16], const float intrinsics[4]);
/* visualisationEngine->CountVisibleBlocks(scene, renderState, minBlockId, maxBlockId)
 * (mapManager->countVisibleBlocks, DenseSlam.cpp:555-556) */
int dslam_count_visible_blocks(dslam_engine *e, const dslam_scene *s, const dslam_render_state *r,
                               int min_block_id, int max_block_id, int *out_count);
/* visualisationEngine->CreateExpectedDepths(scene, pose, intrinsics, renderState) */
int dslam_create_expected_depths(dslam_engine *e, const dslam_scene *s, dslam_render_state *r,
                                 const float M[16], const float intrinsics[4]);
/* visualisationEngine->RenderImage(scene, pose, intrinsics, renderState, outputImage, type): raycast
 * with the render state's range image, then shade.  Exactly one of out_rgba_host (Vector4u per pixel)
 * / out_float_host (float per pixel, DSLAM_IMAGE_DEPTH only) may be non-NULL; both NULL leaves the
 * result on the device (dslam_render_state_image_dev). */
int dslam_render_image(dslam_engine *e, const dslam_scene *s, dslam_render_state *r, const float M[16],
                       const float intrinsics[4], int image_type, uint8_t *out_rgba_host,
                       float *out_float_host);
/* ITMMainEngine::GetImage(out, outFloat, type, pose, intrinsics, localMap) for the FREECAMERA_* types
 * (InfiniTamDriver.cpp:229-277): FindVisibleBlocks + CreateExpectedDepths + RenderImage.
 * Two things the caller gets for free: (1) the render state remembers the view (map version, pose, intrinsics) its
 * raycast result belongs to, so a second image type of the same view -- the GUI's depth + colour pair per tick,
 * DenseSlam.h:146-164 -- is only shaded; any call that can change the map or the render state drops that memo (maps
 * over caller-owned voxel memory are never memoised); (2) an output image inside a dslam_host_alloc buffer is written
 * by the render kernel itself instead of by a copy queued behind it. */
int dslam_get_image(dslam_engine *e, const dslam_scene *s, dslam_render_state *r, const float M[16],
                    const float intrinsics[4], int image_type, uint8_t *out_rgba_host,
                    float *out_float_host);
/* trackingController->Prepare(trackingState, scene, view, renderState) (InfiniTamDriver.h:208-220):
 * CreateExpectedDepths + CreateICPMaps from the render state's own visible list.  Outputs are
 * Vector4f per pixel (points: metres, world frame, w = 1 or -1; normals: w = 0 or -1); may be NULL. */
int dslam_create_icp_maps(dslam_engine *e, const dslam_scene *s, dslam_render_state *r, const float M[16],
                          const float intrinsics[4], float *out_points_host, float *out_normals_host);
/* ITMMainEngine::GetImage(out, NULL, InfiniTAM_IMAGE_SCENERAYCAST, ...) = PreviewType::kRaycastImage
 * (InfiniTamDriver.cpp:28-29), the GUI's default rgb preview (DenseSLAMGUI.h:240): a copy of renderState->raycastImage,
 * the grey rendering upstream's CreateICPMaps draws with the ICP maps (processPixelICP -> drawPixelGrey: all four
 * channels (uchar)((0.8 angle + 0.2) 255) with the image-space normal, 0 where the ray found nothing), i.e. what
 * dslam_create_icp_maps (PrepareNextStepLocalMap, InfiniTamDriver.h:208-220) last left in this render state.
 * Vector4u per pixel.  DSLAM_ERR_INVALID before the first dslam_create_icp_maps (upstream: a zero image -- the
 * mirror's GetImage returns that). */
int dslam_download_raycast_image(dslam_engine *e, const dslam_render_state *r, uint8_t *out_rgba_host);

/* ---- meshing export ---------------------------------------------------------------------------- */
/* ITMMainEngine::SaveCurrSceneToMesh(objFileName, scene) -> ITMMeshingEngine::MeshScene(mesh, scene) (DenseSlam.cpp:
 * 638-643): marching cubes over every allocated voxel block, in upstream's CPU-engine order (hash
 * entries ascending, voxels z/y/x, triangles in case-table order), so the output is deterministic.  A cube is
 * skipped when one of its 8 corner voxels is missing or has sdf == 1 (findPointNeighbors).  max_triangles <= 0
 * selects ITMMesh's noMaxTriangles = num_local_blocks * 32; like upstream the list saturates at max_triangles - 1.
 * The mesh stays on the device until dslam_mesh_download: per triangle 3 vertices x (x, y, z) floats in metres
 * (world frame), and with with_colour the voxel colours interpolated to the same crossings, as (r, g, b) floats
 * in [0, 1] (the coloured-OBJ form of the DynSLAM lineage; upstream v2 writes positions only). */
int dslam_mesh_scene(dslam_engine *e, const dslam_scene *s, int max_triangles, int with_colour,
                     int *out_num_triangles);
int dslam_mesh_download(dslam_engine *e, float *out_positions_host, float *out_colours_host,
                        int capacity_triangles);

/* ---- depth tracker (ICP) ---------------------------------------------------------------------- */
/* trackingController->Track(trackingState, view) (InfiniTamDriver.h:151-163, reached through
 * DenseSlam.cpp:200-206 when the reference runs without ORB-SLAM2 odometry): upstream InfiniTAM v2's
 * ITMDepthTracker::TrackCamera -- point-to-plane ICP of the view's depth image against the points / normals maps
 * that dslam_create_icp_maps left in the render state, coarse to fine over a depth-image pyramid
 * (FilterSubsampleWithHoles), Levenberg-Marquardt damping, 3x3 / 6x6 Cholesky steps.
 * iteration types as upstream's TrackerIterationType. */
enum { DSLAM_TRACKER_ITERATION_ROTATION = 1, DSLAM_TRACKER_ITERATION_TRANSLATION = 2,
       DSLAM_TRACKER_ITERATION_BOTH = 3, DSLAM_TRACKER_ITERATION_NONE = 4 };
#define DSLAM_TRACKER_MAX_LEVELS 8
typedef struct {
  int32_t no_hierarchy_levels;     /* ITMLibSettings::noHierarchyLevels (upstream default 5) */
  int32_t no_icp_run_till_level;   /* ITMLibSettings::noICPRunTillLevel (0) */
  float dist_thresh;               /* depthTrackerICPThreshold (0.1 * 0.1) */
  float termination_threshold;     /* depthTrackerTerminationThreshold (1e-3) */
  int32_t regime[DSLAM_TRACKER_MAX_LEVELS]; /* trackingRegime per level, level 0 = full resolution
                                     * (upstream default: BOTH, BOTH, ROTATION, ROTATION, ROTATION) */
} dslam_tracker_params;
typedef struct {
  int32_t iterations;              /* ComputeGandH evaluations */
  int32_t valid_points_last;       /* noValidPoints of the last evaluation */
  float f_last;                    /* its error value */
  int32_t pad;
} dslam_tracker_result;
/* scene_pose_M = trackingState->pose_pointCloud->GetM() (the pose the ICP maps were rendered from); pose_M is
 * trackingState->pose_d->GetM() on entry and the tracked pose on return.  The view must have been updated. */
int dslam_track_camera(dslam_engine *e, const dslam_view *v, dslam_render_state *r, const float scene_pose_M[16],
                       float pose_M[16], const float intrinsics_d[4], const dslam_tracker_params *params,
                       dslam_tracker_result *result);

/* ---- state read-back (stats for the driver; bulk downloads for parity tests and checkpoints) ------ */
int dslam_get_stats(dslam_engine *e, const dslam_scene *s, const dslam_render_state *r, dslam_stats *out);
int dslam_download_hash_table(dslam_engine *e, const dslam_scene *s, dslam_hash_entry *out_host);
int dslam_download_voxel_blocks(dslam_engine *e, const dslam_scene *s, int first_block, int num_blocks,
                                dslam_voxel *out_host);
int dslam_download_allocation_list(dslam_engine *e, const dslam_scene *s, int32_t *out_host);
int dslam_download_excess_list(dslam_engine *e, const dslam_scene *s, int32_t *out_host);
int dslam_download_visible_ids(dslam_engine *e, const dslam_render_state *r, int32_t *out_host,
                               int capacity, int *out_count);
int dslam_download_visible_types(dslam_engine *e, const dslam_render_state *r, uint8_t *out_host);
int dslam_download_range_image(dslam_engine *e, const dslam_render_state *r, float *out_minmax_host);
int dslam_download_raycast_result(dslam_engine *e, const dslam_render_state *r, float *out_xyzw_host);
int dslam_download_view_depth(dslam_engine *e, const dslam_view *v, float *out_host);
/* the points / normals maps dslam_create_icp_maps left on the device (either pointer may be NULL) */
int dslam_download_icp_maps(dslam_engine *e, const dslam_render_state *r, float *out_points_host,
                            float *out_normals_host);
int dslam_download_swap_states(dslam_engine *e, const dslam_scene *s, uint8_t *out_host);
/* per voxel-block slot: newest global list index that holds the block (-1 never, <= -2 swept by Decay) */
int dslam_download_last_seen(dslam_engine *e, const dslam_scene *s, int32_t *out_host);
/* ITMGlobalCache::GetStoredVoxelBlock(entry): copies the host-stored block; returns 1 if the entry has
 * stored data, 0 if not */
int dslam_download_stored_block(dslam_engine *e, const dslam_scene *s, int entry, dslam_voxel *out_host);
/* scratch of the last allocation pass (entriesAllocType / blockCoords), for bit-exactness tests */
int dslam_download_alloc_scratch(dslam_engine *e, const dslam_scene *s, uint8_t *alloc_types_host,
                                 int16_t *block_coords_host);
/* upload a complete map state (hash table, pool free lists, voxel blocks): checkpoint restore and the
 * stress/roofline generator.  Any pointer may be NULL to keep that part. */
int dslam_upload_scene_state(dslam_engine *e, dslam_scene *s, const dslam_hash_entry *hash_host,
                             const int32_t *allocation_list_host, int last_free_block_id,
                             const int32_t *excess_list_host, int last_free_excess_id);
int dslam_upload_voxel_blocks(dslam_engine *e, dslam_scene *s, int first_block, int num_blocks,
                              const dslam_voxel *host);
int dslam_upload_visible_ids(dslam_engine *e, dslam_render_state *r, const int32_t *ids_host, int count);

/* raw device pointers (HBM layout is documented in DESIGN.md).
 * The hash table is READ-ONLY through this pointer: every pass that lists entries (visibility, decay, window, swapping,
 * meshing, FindVisibleBlocks) walks a bitmap that mirrors `ptr >= 0` of the table (alloc_bits) instead of the table, so an
 * entry written behind the engine's back exists but is never seen.  A caller that must patch or restore the table in
 * place calls dslam_scene_table_changed afterwards (the bitmap is rebuilt from the table: the one pass that reads all of
 * it); dslam_upload_scene_state does that by itself.  Voxel blocks may be written freely (they have no shadow). */
void *dslam_scene_voxel_blocks_dev(dslam_scene *s);
void *dslam_scene_hash_table_dev(dslam_scene *s);
int dslam_scene_table_changed(dslam_engine *e, dslam_scene *s);
void *dslam_render_state_image_dev(dslam_render_state *r, int want_float);

/* ---- sharded re-integration (multi-GPU) ------------------------------------------------ */
/* Restrict the voxel-writing kernels (integrate / de-integrate) of this scene to the voxel-block slots
 * whose chunk (slot / chunk_blocks) satisfies chunk % num_shards == shard; allocation stays global
 * (and bit-identical on every rank).  num_shards = 1 disables sharding.  A scene with host swapping cannot be sharded
 * (DSLAM_ERR_INVALID): every rank would swap its own, partly stale copies out to its own host store, which the block
 * exchange does not cover.  (The reference never turns swapping on: its ITMLibSettings is default-constructed.) */
int dslam_scene_set_shard(dslam_scene *s, int shard, int num_shards, int chunk_blocks);
/* Same, with a contiguous slot range [first_block, first_block + num_blocks): the layout an in-place RCCL
 * all-gather over the voxel-block array needs.  num_blocks < 0 disables. */
int dslam_scene_set_shard_range(dslam_scene *s, int first_block, int num_blocks);
/* The exchange for a batch whose blocks may sit ANYWHERE in the pool (after decay, the sliding window or swapping have
 * returned slots to the free list in arbitrary order, the used slots are no longer a top range): move exactly the blocks
 * the batch touched.  Allocation is replicated and bit-identical on every rank, and the (de-)integration kernels mark
 * every visible resident block they walk over BEFORE their shard test, so all ranks hold the same marks and derive the
 * same per-shard lists of dirty slots -- no ids travel, no counts are exchanged:
 *   dslam_scene_track_dirty(e, s, 1)            clear the marks, start marking         (before the batch)
 *   ... the batch: dslam_deprocess_frame / dslam_process_frame under dslam_scene_set_shard(rank, world, chunk) ...
 *   dslam_shard_dirty_plan(e, s, world, chunk, counts)   counts[r] = dirty blocks of shard r, identical on all ranks
 *   dslam_shard_dirty_pack(e, s, rank, send, cap)        this rank's dirty blocks, ascending in slot, 4096 B each
 *   ncclAllGather(send, recv, cap * 4096 bytes) with cap = max(counts)   (the one collective)
 *   dslam_shard_dirty_unpack(e, s, rank, recv, cap)      every other shard's blocks into place
 *   dslam_scene_track_dirty(e, s, 0)
 * num_local_blocks must be a multiple of world * chunk; world <= 64. */
int dslam_scene_track_dirty(dslam_engine *e, dslam_scene *s, int enable);
int dslam_shard_dirty_plan(dslam_engine *e, dslam_scene *s, int num_shards, int chunk_blocks, int32_t *counts_out);
int dslam_shard_dirty_pack(dslam_engine *e, const dslam_scene *s, int shard, void *send_dev, int capacity_blocks);
int dslam_shard_dirty_unpack(dslam_engine *e, dslam_scene *s, int skip_shard, const void *recv_dev, int stride_blocks);

/* ---- instrumentation ---------------------------------------------------------------------------- */
/* Time `iterations` back-to-back launches of the integrate kernel alone on the engine stream with HIP
 * events (state is restored afterwards is NOT guaranteed: use on a scratch scene).  Returns the mean
 * milliseconds per launch and the number of visible blocks processed per launch. */
int dslam_time_integrate(dslam_engine *e, dslam_scene *s, const dslam_view *v, const dslam_render_state *r,
                         const float M_d[16], const float intrinsics_d[4], int iterations,
                         float *out_ms_per_launch, int *out_visible_blocks);
/* accumulated HIP-event time of the integrate kernel launched by dslam_process_frame /
 * dslam_integrate_into_scene since the last reset (events are recorded only while enabled). */
int dslam_kernel_timer_enable(dslam_engine *e, int enable);
int dslam_kernel_timer_read(dslam_engine *e, double *out_integrate_ms, int64_t *out_launches,
                            int64_t *out_visible_blocks);

#ifdef __cplusplus
}
#endif
#endif /* DSLAM_FUSION_H */
