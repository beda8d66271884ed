// ITMMainEngine.h -- the ITMLib engine classes InfiniTamDriver derives from and calls (reference include at
// InfiniTamDriver.h:11), implemented as thin host wrappers over the C ABI of libdslam_fusion.so.
//
//   ITMMainEngine(settings, calib, imgSize_rgb, imgSize_d)            InfiniTamDriver.h:102
//   members view, settings, denseMapper, trackingController, viewBuilder, visualisationEngine, mapManager,
//           mActiveDataManger                                         InfiniTamDriver.h:133-157,189-241,276-308,363
//   GetPrimaryLocalMap(), GetImage(out, outFloat, type, pose, intrinsics, localMap), SaveCurrSceneToMesh
//                                                                      InfiniTamDriver.h:146,169; InfiniTamDriver.cpp:242-275
//
// How calls complete.  The reference's driver treats every ITMLib call as done when it returns, but it only ever LOOKS at
// results in a few places: the image GetImage filled (InfiniTamDriver.cpp:242-250,269-277), the host mirrors view->rgb /
// view->depth (InfiniTamDriver.h:217-218), the counters lastFreeBlockId / noVisibleEntries / GetDecayedBlockCount
// (InfiniTamDriver.h:210,345,368), the tracked pose (:157-161), the mesh file (DenseSlam.cpp:641).  Between UpdateView,
// IntegrateLocalMap, SlideWindow* and Decay (DenseSlam.cpp:210-232) it looks at nothing.  So the mirror runs the engine
// asynchronously (dslam_engine_set_async): ProcessFrame, DeProcessFrame, Decay*, SlideWindow*, ResetScene, Prepare, the
// swapping calls and the visualisation steps ENQUEUE their kernels and return; UpdateView waits only for its own upload
// (which runs on the copy stream under the kernels enqueued before it; the caller may rewrite its images as soon as
// it returns); everything that hands data to the host WAITS: GetImage, a counter read, GetData on a stale host mirror,
// Track (per ICP iteration), MeshScene, CountVisibleBlocks.  What the device can only report late (dslam_fusion.h,
// "conditions only the device can detect") is thrown by the first of those waiting calls.  A caller that observes
// nothing between two waiting calls cannot tell the difference from the synchronous engine: same kernels, same order,
// same stream.  DSLAM_MIRROR_SYNC=1 in the environment gives the synchronous engine back (every call waits), for A/B
// measurements and debugging.
// Header-only; link with -ldslam_fusion.
#pragma once
#include <cstdio>
#include <vector>

#include "../Objects/ITMObjects.h"

namespace ITMLib {
namespace Engine {
using namespace Objects;

/// ITMLib::Engine::WeightParams (SystemEntry.cpp:183-187; InfiniTamDriver.h:101,391)
struct WeightParams {
  bool depthWeighting;
  int maxNewW;
  float maxDistance;
  WeightParams() : depthWeighting(false), maxNewW(1), maxDistance(1.0f) {}
};

/// ITMSwappingEngine<TVoxel,TIndex>: SaveToGlobalMemory(scene) is Hansry's one-argument form (DenseSlam.h:250)
template <class TVoxel, class TIndex> class ITMSwappingEngine {
  dslam_engine *eng_;
 public:
  explicit ITMSwappingEngine(dslam_engine *e) : eng_(e) {}
  void IntegrateGlobalIntoLocal(ITMScene<TVoxel, TIndex> *scene, ITMRenderState *rs) {
    dslam_check(dslam_swap_in(eng_, scene->handle, rs ? rs->handle : nullptr), "dslam_swap_in");
    scene->MarkCountersStale(rs);
  }
  void SaveToGlobalMemory(ITMScene<TVoxel, TIndex> *scene, ITMRenderState *rs) {
    dslam_check(dslam_swap_out(eng_, scene->handle, rs->handle), "dslam_swap_out");
    scene->MarkCountersStale(rs);
  }
  void SaveToGlobalMemory(ITMScene<TVoxel, TIndex> *scene) {
    dslam_check(dslam_save_to_global_memory(eng_, scene->handle), "dslam_save_to_global_memory");
    scene->MarkCountersStale();
  }
};

/// ITMDenseMapper: the fusion entry points (InfiniTamDriver.h:187-199,241,274-310,354-370)
class ITMDenseMapper {
  dslam_engine *eng_;
  const ITMRGBDCalib *calib_;
  ITMSwappingEngine<ITMVoxel, ITMVoxelIndex> *swappingEngine_;
  // the scenes this mapper has worked on (GetDecayedBlockCount sums their device-resident counters when asked; scenes
  // belong to the map manager, which lives as long as the engine that owns this mapper)
  std::vector<const ITMScene<ITMVoxel, ITMVoxelIndex> *> scenes_;
  void note(const ITMScene<ITMVoxel, ITMVoxelIndex> *scene) {
    for (size_t i = 0; i < scenes_.size(); i++) if (scenes_[i] == scene) return;
    scenes_.push_back(scene);
  }

  void poseArgs(const ITMView *view, const ITMTrackingState *ts, Matrix4f &M_d, Matrix4f &M_rgb, Vector4f &kd, Vector4f &kr) const {
    M_d = ts->pose_d->GetM();
    M_rgb = view->calib->trafo_rgb_to_depth.calib_inv * M_d;
    kd = view->calib->intrinsics_d.projectionParamsSimple.all;
    kr = view->calib->intrinsics_rgb.projectionParamsSimple.all;
  }

 public:
  ITMDenseMapper(dslam_engine *e, const ITMRGBDCalib *calib)
      : eng_(e), calib_(calib), swappingEngine_(new ITMSwappingEngine<ITMVoxel, ITMVoxelIndex>(e)) {}
  ~ITMDenseMapper() { delete swappingEngine_; }

  void SetFusionWeightParams(const WeightParams &p) {
    dslam_weight_params w = {p.depthWeighting ? 1 : 0, p.maxNewW, p.maxDistance};
    dslam_check(dslam_set_fusion_weight_params(eng_, &w), "dslam_set_fusion_weight_params");
  }
  void ResetScene(ITMScene<ITMVoxel, ITMVoxelIndex> *scene) {
    dslam_check(dslam_scene_reset(eng_, scene->handle), "dslam_scene_reset");
    scene->MarkCountersStale();
    note(scene);
  }
  void ProcessFrame(const ITMView *view, const ITMTrackingState *ts, ITMScene<ITMVoxel, ITMVoxelIndex> *scene,
                    ITMRenderState *rs, bool onlyUpdateVisibleList = false, bool isDefusion = false) {
    Matrix4f M_d, M_rgb; Vector4f kd, kr;
    poseArgs(view, ts, M_d, M_rgb, kd, kr);
    dslam_check(dslam_process_frame(eng_, scene->handle, view->handle, rs->handle, M_d.m, kd.v, M_rgb.m, kr.v,
                                    onlyUpdateVisibleList, isDefusion), "dslam_process_frame");
    scene->MarkCountersStale(rs);
    note(scene);
  }
  void DeProcessFrame(const ITMView *view, const ITMTrackingState *ts, ITMScene<ITMVoxel, ITMVoxelIndex> *scene,
                      ITMRenderState *rs) {
    Matrix4f M_d, M_rgb; Vector4f kd, kr;
    poseArgs(view, ts, M_d, M_rgb, kd, kr);
    dslam_check(dslam_deprocess_frame(eng_, scene->handle, view->handle, rs->handle, M_d.m, kd.v, M_rgb.m, kr.v),
                "dslam_deprocess_frame");
    scene->MarkCountersStale(rs);
    note(scene);
  }
  /// (extension of this engine, not in upstream) keep the visible list of the fusion that has just run with the keyframe's
  /// images in the device-resident store: what ReProcessFrames de-integrates from (dslam_frame_store_put_visible_list)
  void KeepVisibleList(dslam_frame_store *store, int slot, ITMScene<ITMVoxel, ITMVoxelIndex> *scene, ITMRenderState *rs) {
    dslam_check(dslam_frame_store_put_visible_list(eng_, store, slot, scene->handle, rs->handle), "dslam_frame_store_put_visible_list");
  }
  /// (extension) the re-fusion loop of DenseSlam::OnlineCorrection [REF DenseSlam.cpp:389-403] -- per keyframe DeProcessFrame at
  /// the old pose, ProcessFrame(isDefusion) at the new one -- as ONE call, run block-major on the device (dslam_reintegrate_batch;
  /// one camera).  De-integration visits the blocks of the keyframe's own stored list, not what an allocation pass at the old
  /// pose finds today: not the reference's call sequence.
  void ReProcessFrames(const ITMView *view, ITMScene<ITMVoxel, ITMVoxelIndex> *scene, ITMRenderState *rs, dslam_frame_store *store,
                       int n, const int *slots, const Matrix4f *oldM_d, const Matrix4f *newM_d) {
    const Vector4f kd = view->calib->intrinsics_d.projectionParamsSimple.all;
    const Vector2f ab = calib_->disparityCalib.params;
    static_assert(sizeof(Matrix4f) == 16 * sizeof(float), "poses are handed over as n x 16 floats");
    dslam_check(dslam_reintegrate_batch(eng_, scene->handle, view->handle, rs->handle, store, n, slots, n ? oldM_d[0].m : nullptr,
                                        n ? newM_d[0].m : nullptr, kd.v, ab.x, ab.y), "dslam_reintegrate_batch");
    scene->MarkCountersStale(rs);
    note(scene);
  }
  void Decay(ITMScene<ITMVoxel, ITMVoxelIndex> *scene, ITMRenderState *rs, int maxWeight, int minAge, bool forceAllVoxels) {
    dslam_check(dslam_decay(eng_, scene->handle, rs ? rs->handle : nullptr, maxWeight, minAge, forceAllVoxels), "dslam_decay");
    scene->MarkCountersStale(rs);
    note(scene);
  }
  void DecayDefusionPart(ITMScene<ITMVoxel, ITMVoxelIndex> *scene, ITMRenderState *rs, int maxWeight, int minAge, bool forceAllVoxels) {
    dslam_check(dslam_decay_defusion_part(eng_, scene->handle, rs ? rs->handle : nullptr, maxWeight, minAge, forceAllVoxels),
                "dslam_decay_defusion_part");
    scene->MarkCountersStale(rs);
    note(scene);
  }
  void SlideWindow(ITMScene<ITMVoxel, ITMVoxelIndex> *scene, ITMRenderState *rs, int maxAge) {
    dslam_check(dslam_slide_window(eng_, scene->handle, rs ? rs->handle : nullptr, maxAge), "dslam_slide_window");
    scene->MarkCountersStale(rs);
    note(scene);
  }
  void SlideWindowDefusionPart(ITMScene<ITMVoxel, ITMVoxelIndex> *scene, ITMRenderState *rs, int maxAge, int maxSize) {
    dslam_check(dslam_slide_window_defusion_part(eng_, scene->handle, rs ? rs->handle : nullptr, maxAge, maxSize),
                "dslam_slide_window_defusion_part");
    scene->MarkCountersStale(rs);
    note(scene);
  }
  /// InfiniTamDriver.h:366-370 (the GUI's "saved memory" figure): read from the device when asked, not after every call
  size_t GetDecayedBlockCount() const {
    long long n = 0;
    for (size_t i = 0; i < scenes_.size(); i++) n += scenes_[i]->Counters().decayed_block_count;
    return (size_t)n;
  }
  ITMSwappingEngine<ITMVoxel, ITMVoxelIndex> *GetSwappingEngine() { return swappingEngine_; }
};

/// ITMViewBuilder::UpdateView(&view, rgb, rawDepth, timestamp, useBilateralFilter) (InfiniTamDriver.cpp:286)
class ITMViewBuilder {
  dslam_engine *eng_;
  const ITMRGBDCalib *calib_;
 public:
  ITMViewBuilder(dslam_engine *e, const ITMRGBDCalib *c) : eng_(e), calib_(c) {}
  const ITMRGBDCalib *GetCalib() const { return calib_; }  ///< InfiniTamDriver.cpp:241
  void UpdateView(ITMView **view, ITMUChar4Image *rgb, ITMShortImage *rawDepth, double timestamp, bool useBilateralFilter) {
    if (*view == nullptr) *view = new ITMView(calib_, rgb->noDims, rawDepth->noDims, eng_);
    ITMView *v = *view;
    const Vector2f ab = calib_->disparityCalib.params;
    dslam_check(dslam_view_update(eng_, v->handle, &rgb->GetData(MEMORYDEVICE_CPU)->x, rawDepth->GetData(MEMORYDEVICE_CPU),
                                  ab.x, ab.y, timestamp, useBilateralFilter), "dslam_view_update");
    v->timestamp = timestamp;
    // host mirrors: view->rgb / view->depth are read back by the driver (InfiniTamDriver.h:217-218) -- on demand
    v->MarkHostStale();
  }
  /// UpdateView for a keyframe that already sits in the device-resident keyframe store (the UpdateView calls of
  /// DenseSlam::OnlineCorrection, DenseSlam.cpp:392,421): no upload; the host mirrors view->rgb / view->depth are
  /// refreshed only if somebody reads them (nothing on that path does).
  void UpdateViewFromStore(ITMView **view, const dslam_frame_store *store, int slot, double timestamp, bool useBilateralFilter) {
    if (*view == nullptr) throw std::runtime_error("UpdateViewFromStore: the view must have been created by UpdateView first");
    const Vector2f ab = calib_->disparityCalib.params;
    dslam_check(dslam_view_update_from_store(eng_, (*view)->handle, store, slot, ab.x, ab.y, timestamp, useBilateralFilter),
                "dslam_view_update_from_store");
    (*view)->timestamp = timestamp;
    (*view)->MarkHostStale();
  }
};

/// ITMTrackingController: Prepare = raycast into ICP maps (InfiniTamDriver.h:212-215); Track = ITMDepthTracker's
/// point-to-plane ICP of the view against those maps (InfiniTamDriver.h:151-163), both on the device.
class ITMTrackingController {
  dslam_engine *eng_;
  dslam_tracker_params params_;
 public:
  explicit ITMTrackingController(dslam_engine *e, const ITMLibSettings *settings = nullptr) : eng_(e) {
    ITMLibSettings defaults;
    if (settings == nullptr) settings = &defaults;
    params_.no_hierarchy_levels = settings->noHierarchyLevels;
    params_.no_icp_run_till_level = settings->noICPRunTillLevel;
    params_.dist_thresh = settings->depthTrackerICPThreshold;
    params_.termination_threshold = settings->depthTrackerTerminationThreshold;
    for (int i = 0; i < DSLAM_TRACKER_MAX_LEVELS; i++) params_.regime[i] = (int)settings->trackingRegime[i];
  }
  void Prepare(ITMTrackingState *ts, const ITMScene<ITMVoxel, ITMVoxelIndex> *scene, const ITMView *view, ITMRenderState *rs) {
    const Matrix4f M = ts->pose_d->GetM();
    const Vector4f k = view->calib->intrinsics_d.projectionParamsSimple.all;
    // the maps stay on the device (Track reads them there); trackingState->pointsMap / normalsMap are host mirrors
    // filled when somebody asks for them
    dslam_check(dslam_create_icp_maps(eng_, scene->handle, rs->handle, M.m, k.v, nullptr, nullptr), "dslam_create_icp_maps");
    dslam_engine *eng = eng_;
    dslam_render_state *h = rs->handle;
    ts->pointsMap->SetHostPull([eng, h](Vector4f *dst) { dslam_check(dslam_download_icp_maps(eng, h, &dst->x, nullptr), "dslam_download_icp_maps"); });
    ts->normalsMap->SetHostPull([eng, h](Vector4f *dst) { dslam_check(dslam_download_icp_maps(eng, h, nullptr, &dst->x), "dslam_download_icp_maps"); });
    ts->pointsMap->MarkHostStale();
    ts->normalsMap->MarkHostStale();
    // renderState->raycastImage: the grey rendering CreateICPMaps draws next to the maps (InfiniTAM_IMAGE_SCENERAYCAST)
    rs->raycastImage->SetHostPull([eng, h](Vector4u *dst) { dslam_check(dslam_download_raycast_image(eng, h, &dst->x), "dslam_download_raycast_image"); });
    rs->raycastImage->MarkHostStale();
    ts->pose_pointCloud->SetM(M);
    ts->preparedWith = rs;
    ts->age_pointCloud = 0;
  }
  void Track(ITMTrackingState *ts, const ITMView *view) {
    if (ts->preparedWith == nullptr) throw std::runtime_error("ITMTrackingController::Track: Prepare has not been called");
    const Vector4f k = view->calib->intrinsics_d.projectionParamsSimple.all;
    Matrix4f M = ts->pose_d->GetM();
    dslam_tracker_result res;
    dslam_check(dslam_track_camera(eng_, view->handle, ts->preparedWith->handle, ts->pose_pointCloud->GetM().m, M.m, k.v,
                                   &params_, &res), "dslam_track_camera");
    ts->pose_d->SetM(M);
    ts->age_pointCloud++;
  }
};

/// ITMVisualisationEngine (InfiniTamDriver.h:362-364)
template <class TVoxel, class TIndex> class ITMVisualisationEngine {
  dslam_engine *eng_;
 public:
  explicit ITMVisualisationEngine(dslam_engine *e) : eng_(e) {}
  ITMRenderState *CreateRenderState(const ITMScene<TVoxel, TIndex> *scene, Vector2i sz) const { return new ITMRenderState_VH(eng_, scene->handle, sz); }
  void FindVisibleBlocks(const ITMScene<TVoxel, TIndex> *scene, const ITMPose *pose, const ITMIntrinsics *intr, ITMRenderState *rs) const {
    dslam_check(dslam_find_visible_blocks(eng_, scene->handle, rs->handle, pose->GetM().m, intr->projectionParamsSimple.all.v), "dslam_find_visible_blocks");
    rs->MarkCountersStale();
  }
  int CountVisibleBlocks(const ITMScene<TVoxel, TIndex> *scene, const ITMRenderState *rs, int minBlockId, int maxBlockId) const {
    int n = 0;
    dslam_check(dslam_count_visible_blocks(eng_, scene->handle, rs->handle, minBlockId, maxBlockId, &n), "dslam_count_visible_blocks");
    return n;
  }
  void CreateExpectedDepths(const ITMScene<TVoxel, TIndex> *scene, const ITMPose *pose, const ITMIntrinsics *intr, ITMRenderState *rs) const {
    dslam_check(dslam_create_expected_depths(eng_, scene->handle, rs->handle, pose->GetM().m, intr->projectionParamsSimple.all.v), "dslam_create_expected_depths");
  }
};

/// ITMVoxelMapGraphManager (DenseSlam.cpp:135-152,555-556; InfiniTamDriver.h:136,269): host container of local maps
class ITMVoxelMapGraphManager {
  const ITMLibSettings *settings_;
  dslam_engine *eng_;
  Vector2i trackedImageSize_;
  const ITMVisualisationEngine<ITMVoxel, ITMVoxelIndex> *vis_;
  std::vector<ITMLocalMap *> maps_;
 public:
  ITMVoxelMapGraphManager(const ITMLibSettings *s, dslam_engine *e, const ITMVisualisationEngine<ITMVoxel, ITMVoxelIndex> *vis, Vector2i sz)
      : settings_(s), eng_(e), trackedImageSize_(sz), vis_(vis) {}
  ~ITMVoxelMapGraphManager() { for (auto *m : maps_) delete m; }
  int createNewLocalMap() { maps_.push_back(new ITMLocalMap(settings_, eng_, trackedImageSize_)); return (int)maps_.size() - 1; }
  int numLocalMaps() const { return (int)maps_.size(); }
  ITMLocalMap *getLocalMap(int i) const { return (i < 0 || i >= (int)maps_.size()) ? nullptr : maps_[i]; }
  void setEstimatedGlobalPose(int i, const ITMPose &pose) { maps_[i]->estimatedGlobalPose = pose; }
  int getLocalMapSize(int i) const {
    dslam_stats st;
    dslam_check(dslam_get_stats(eng_, maps_[i]->scene->handle, nullptr, &st), "dslam_get_stats");
    return st.num_allocated_blocks - st.last_free_block_id - 1;
  }
  int countVisibleBlocks(int i, int minBlockId, int maxBlockId, bool /*invertIDs*/) const {
    return vis_->CountVisibleBlocks(maps_[i]->scene, maps_[i]->renderState, minBlockId, maxBlockId);
  }
};

/// ITMActiveMapManager: only numActiveLocalMaps() is reached (InfiniTamDriver.h:264)
class ITMActiveMapManager {
  ITMVoxelMapGraphManager *maps_;
 public:
  explicit ITMActiveMapManager(ITMVoxelMapGraphManager *m) : maps_(m) {}
  int numActiveLocalMaps() const { return maps_->numLocalMaps() > 0 ? 1 : 0; }
};

class ITMMainEngine {
 public:
  enum GetImageType {
    InfiniTAM_IMAGE_ORIGINAL_RGB,
    InfiniTAM_IMAGE_ORIGINAL_DEPTH,
    InfiniTAM_IMAGE_SCENERAYCAST,
    InfiniTAM_IMAGE_FREECAMERA_SHADED,
    InfiniTAM_IMAGE_FREECAMERA_COLOUR_FROM_VOLUME,
    InfiniTAM_IMAGE_FREECAMERA_COLOUR_FROM_NORMAL,
    InfiniTAM_IMAGE_FREECAMERA_DEPTH,
    InfiniTAM_IMAGE_UNKNOWN
  };

  ITMMainEngine(const ITMLibSettings *settings_, const ITMRGBDCalib *calib, Vector2i imgSize_rgb, Vector2i imgSize_d = Vector2i(-1, -1))
      : settings(settings_), view(nullptr), engine_(nullptr), freeviewScene_(nullptr), renderState_freeview_(nullptr) {
    if (imgSize_d.x == -1 || imgSize_d.y == -1) imgSize_d = imgSize_rgb;
    dslam_check(dslam_engine_create(settings->hipDeviceIndex, &engine_), "dslam_engine_create");
    // calls enqueue; the calls that hand data to the host wait (see the head of this file)
    const char *sync_env = getenv("DSLAM_MIRROR_SYNC");
    deferred_ = !(sync_env && atoi(sync_env) != 0);
    dslam_check(dslam_engine_set_async(engine_, deferred_ ? 1 : 0), "dslam_engine_set_async");
    denseMapper = new ITMDenseMapper(engine_, calib);
    viewBuilder = new ITMViewBuilder(engine_, calib);
    trackingController = new ITMTrackingController(engine_, settings);
    visualisationEngine = new ITMVisualisationEngine<ITMVoxel, ITMVoxelIndex>(engine_);
    mapManager = new ITMVoxelMapGraphManager(settings, engine_, visualisationEngine, imgSize_d);
    mActiveDataManger = new ITMActiveMapManager(mapManager);
  }
  virtual ~ITMMainEngine() {
    delete renderState_freeview_;
    delete mActiveDataManger; delete mapManager; delete visualisationEngine; delete trackingController;
    delete viewBuilder; delete denseMapper; delete view;
    dslam_engine_destroy(engine_);
  }

  ITMLocalMap *GetPrimaryLocalMap() const { return mapManager->getLocalMap(0); }

  /// FREECAMERA_* render the given local map (primary when null; nothing before the first keyframe, B.1);
  /// SCENERAYCAST copies renderState->raycastImage, the grey tracking raycast trackingController->Prepare draws
  /// (zero until the first Prepare, as upstream); ORIGINAL_* copy the view.
  void GetImage(ITMUChar4Image *out, ITMFloatImage *outFloat, GetImageType type, ITMPose *pose = nullptr,
                ITMIntrinsics *intrinsics = nullptr, const ITMLocalMap *localMap = nullptr) {
    if (view == nullptr) return;
    if (localMap == nullptr) localMap = GetPrimaryLocalMap();
    switch (type) {
      case InfiniTAM_IMAGE_ORIGINAL_RGB:
        if (out) { out->ChangeDims(view->rgb->noDims); memcpy(out->GetData(MEMORYDEVICE_CPU), view->rgb->GetData(MEMORYDEVICE_CPU), out->dataSize * 4); }
        return;
      case InfiniTAM_IMAGE_ORIGINAL_DEPTH:
        if (outFloat) { outFloat->ChangeDims(view->depth->noDims); memcpy(outFloat->GetData(MEMORYDEVICE_CPU), view->depth->GetData(MEMORYDEVICE_CPU), outFloat->dataSize * 4); }
        return;
      case InfiniTAM_IMAGE_SCENERAYCAST:
        if (out && localMap && localMap->renderState->raycastImage) {
          out->ChangeDims(localMap->renderState->raycastImage->noDims);
          memcpy(out->GetData(MEMORYDEVICE_CPU), localMap->renderState->raycastImage->GetData(MEMORYDEVICE_CPU), out->dataSize * 4);
        }
        return;
      default: break;
    }
    if (localMap == nullptr || pose == nullptr || intrinsics == nullptr) return;
    int t;
    switch (type) {
      case InfiniTAM_IMAGE_FREECAMERA_SHADED: t = DSLAM_IMAGE_SHADED; break;
      case InfiniTAM_IMAGE_FREECAMERA_COLOUR_FROM_VOLUME: t = DSLAM_IMAGE_COLOUR_FROM_VOLUME; break;
      case InfiniTAM_IMAGE_FREECAMERA_COLOUR_FROM_NORMAL: t = DSLAM_IMAGE_COLOUR_FROM_NORMAL; break;
      case InfiniTAM_IMAGE_FREECAMERA_DEPTH: t = DSLAM_IMAGE_DEPTH; break;
      default: return;
    }
    const Vector2i sz = (t == DSLAM_IMAGE_DEPTH) ? (outFloat ? outFloat->noDims : Vector2i(0, 0)) : (out ? out->noDims : Vector2i(0, 0));
    if (sz.x <= 0) return;
    if (renderState_freeview_ == nullptr || freeviewScene_ != localMap->scene || freeviewSize_.x != sz.x || freeviewSize_.y != sz.y) {
      delete renderState_freeview_;
      renderState_freeview_ = visualisationEngine->CreateRenderState(localMap->scene, sz);
      freeviewScene_ = localMap->scene; freeviewSize_ = sz;
    }
    dslam_check(dslam_get_image(engine_, localMap->scene->handle, renderState_freeview_->handle, pose->GetM().m,
                                intrinsics->projectionParamsSimple.all.v, t,
                                t == DSLAM_IMAGE_DEPTH ? nullptr : &out->GetData(MEMORYDEVICE_CPU)->x,
                                t == DSLAM_IMAGE_DEPTH ? outFloat->GetData(MEMORYDEVICE_CPU) : nullptr), "dslam_get_image");
    // the caller reads the image next (ItmToCv / ItmDepthToCv, InfiniTamDriver.cpp:249,276): a page-locked image is filled
    // by the render kernel itself, so this is where the mirror waits for the stream -- and hears what the device reported
    if (deferred_) dslam_check(dslam_engine_synchronize(engine_), "dslam_engine_synchronize");
  }

  /// SaveCurrSceneToMesh(objFileName, scene) (DenseSlam.cpp:641): meshingEngine->MeshScene(mesh, scene), then
  /// mesh->WriteOBJ(objFileName).  The marching cubes run on the device; the triangle list comes back in the order
  /// of upstream's CPU engine, so the file is the same from run to run.
  void SaveCurrSceneToMesh(const char *objFileName, const ITMScene<ITMVoxel, ITMVoxelIndex> *scene) {
    ITMMesh mesh((unsigned)settings->numLocalBlocks * 32u);
    MeshScene(&mesh, scene);
    mesh.WriteOBJ(objFileName);
  }
  /// ITMMeshingEngine::MeshScene(mesh, scene)
  void MeshScene(ITMMesh *mesh, const ITMScene<ITMVoxel, ITMVoxelIndex> *scene) {
    int n = 0;
    dslam_check(dslam_mesh_scene(engine_, scene->handle, (int)mesh->noMaxTriangles, settings->meshWithColour, &n), "dslam_mesh_scene");
    mesh->noTotalTriangles = (unsigned)n;
    mesh->triangles.resize((size_t)n + 1);  // never a null data()
    mesh->colours.resize(settings->meshWithColour ? (size_t)n + 1 : 0);
    dslam_check(dslam_mesh_download(engine_, &mesh->triangles[0].p0.x, settings->meshWithColour ? &mesh->colours[0].p0.x : nullptr, n),
                "dslam_mesh_download");
    mesh->triangles.resize((size_t)n);
    if (settings->meshWithColour) mesh->colours.resize((size_t)n);
  }

  dslam_engine *GetDslamEngine() const { return engine_; }

 protected:
  const ITMLibSettings *settings;
  ITMView *view;
  ITMDenseMapper *denseMapper;
  ITMViewBuilder *viewBuilder;
  ITMTrackingController *trackingController;
  ITMVisualisationEngine<ITMVoxel, ITMVoxelIndex> *visualisationEngine;
  ITMVoxelMapGraphManager *mapManager;
  ITMActiveMapManager *mActiveDataManger;

 private:
  dslam_engine *engine_;
  bool deferred_;   ///< the engine runs asynchronously (default; DSLAM_MIRROR_SYNC=1 turns it off)
  const ITMScene<ITMVoxel, ITMVoxelIndex> *freeviewScene_;
  ITMRenderState *renderState_freeview_;
  Vector2i freeviewSize_;
};

}  // namespace Engine
}  // namespace ITMLib

// global names the reference uses unqualified (InfiniTamDriver.h:132-137,362; DenseSlam.h:511)
using ITMLib::Engine::ITMActiveMapManager;
using ITMLib::Engine::ITMLocalMap;
using ITMLib::Engine::ITMVisualisationEngine;
using ITMLib::Engine::ITMVoxelMapGraphManager;
using ITMLib::Objects::ITMRenderState_VH;
