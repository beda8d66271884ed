// ITMObjects.h -- ITMPose, ITMLibSettings, ITMSceneParams, ITMView, ITMTrackingState, ITMRenderState(_VH), ITMScene,
// ITMLocalMap: the object surface InfiniTamDriver / DenseSlam touch, each holding the C-ABI
// handle of its device-side counterpart.
#pragma once
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "ITMRGBDCalib.h"

namespace ITMLib {
namespace Objects {

/// ITMPose: M = world -> camera.  SetM/SetInvM/GetM/GetInvM/GetParams/Coerce
/// (InfiniTamDriver.h:153,159,174-177; InfiniTamDriver.cpp:47-49; DenseSlam.cpp:139,330-337).
class ITMPose {
  Matrix4f M;
  float params_[6];  // tx, ty, tz, rx, ry, rz (se3 log, DenseSlam.cpp:332-336)

  void paramsFromM() {
    const float R[3][3] = {{M.at(0, 0), M.at(1, 0), M.at(2, 0)}, {M.at(0, 1), M.at(1, 1), M.at(2, 1)}, {M.at(0, 2), M.at(1, 2), M.at(2, 2)}};
    const float t[3] = {M.at(3, 0), M.at(3, 1), M.at(3, 2)};
    // rotation vector as upstream's SetParamsFromModelView structures it [UPSTREAM-RECALL]: asin of the
    // antisymmetric part for small angles (acos loses all precision near 1), acos in the middle range, and the
    // symmetric part for the axis near pi
    const float c = (R[0][0] + R[1][1] + R[2][2] - 1.0f) * 0.5f;
    float w[3] = {(R[2][1] - R[1][2]) * 0.5f, (R[0][2] - R[2][0]) * 0.5f, (R[1][0] - R[0][1]) * 0.5f};
    const float s = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    float angle = 0.0f;
    if (c > 0.70710678f) {
      if (s > 0.0f) { angle = asinf(s > 1.0f ? 1.0f : s); const float k = angle / s; w[0] *= k; w[1] *= k; w[2] *= k; }
    } else if (c > -0.70710678f) {
      angle = acosf(c);
      const float k = angle / s; w[0] *= k; w[1] *= k; w[2] *= k;
    } else {
      angle = 3.14159265358979f - asinf(s > 1.0f ? 1.0f : s);
      const float d[3] = {R[0][0] - c, R[1][1] - c, R[2][2] - c};
      float r2[3];
      if (fabsf(d[0]) > fabsf(d[1]) && fabsf(d[0]) > fabsf(d[2])) { r2[0] = d[0]; r2[1] = (R[1][0] + R[0][1]) * 0.5f; r2[2] = (R[0][2] + R[2][0]) * 0.5f; }
      else if (fabsf(d[1]) > fabsf(d[2])) { r2[0] = (R[1][0] + R[0][1]) * 0.5f; r2[1] = d[1]; r2[2] = (R[2][1] + R[1][2]) * 0.5f; }
      else { r2[0] = (R[0][2] + R[2][0]) * 0.5f; r2[1] = (R[2][1] + R[1][2]) * 0.5f; r2[2] = d[2]; }
      if (r2[0] * w[0] + r2[1] * w[1] + r2[2] * w[2] < 0.0f) { r2[0] = -r2[0]; r2[1] = -r2[1]; r2[2] = -r2[2]; }
      const float n = sqrtf(r2[0] * r2[0] + r2[1] * r2[1] + r2[2] * r2[2]);
      for (int i = 0; i < 3; i++) w[i] = n > 0.0f ? angle * r2[i] / n : 0.0f;
    }
    // t = V u  ->  u = V^-1 t with V = I + B [w]x + C [w]x^2
    float u[3] = {t[0], t[1], t[2]};
    if (angle > 1e-6f) {
      const float a2 = angle * angle;
      // D = (1 - (a/2) cot(a/2)) / a^2: by its series below 0.1 rad, where the closed form cancels catastrophically
      float D;
      if (angle < 0.1f) D = 1.0f / 12.0f + a2 * (1.0f / 720.0f + a2 * (1.0f / 30240.0f));
      else { const float h = 0.5f * angle; D = (1.0f - h * cosf(h) / sinf(h)) / a2; }
      const float wxt[3] = {w[1] * t[2] - w[2] * t[1], w[2] * t[0] - w[0] * t[2], w[0] * t[1] - w[1] * t[0]};
      const float wxwxt[3] = {w[1] * wxt[2] - w[2] * wxt[1], w[2] * wxt[0] - w[0] * wxt[2], w[0] * wxt[1] - w[1] * wxt[0]};
      for (int i = 0; i < 3; i++) u[i] = t[i] - 0.5f * wxt[i] + D * wxwxt[i];
    }
    for (int i = 0; i < 3; i++) { params_[i] = u[i]; params_[3 + i] = w[i]; }
  }

 public:
  ITMPose() { M.setIdentity(); for (float &p : params_) p = 0; }
  void SetM(const Matrix4f &m) { M = m; paramsFromM(); }
  void SetInvM(const Matrix4f &invM) { Matrix4f m; invM.inv(m); SetM(m); }
  const Matrix4f &GetM() const { return M; }
  Matrix4f GetInvM() const { Matrix4f r; M.inv(r); return r; }
  const float *GetParams() const { return params_; }
  /// re-orthonormalise the rotation part (Gram-Schmidt) and force the bottom row to (0,0,0,1)
  void Coerce() {
    float c0[3] = {M.m[0], M.m[1], M.m[2]}, c1[3] = {M.m[4], M.m[5], M.m[6]}, c2[3];
    float n = sqrtf(c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2]);
    if (n > 0) for (float &v : c0) v /= n;
    const float d = c0[0] * c1[0] + c0[1] * c1[1] + c0[2] * c1[2];
    for (int i = 0; i < 3; i++) c1[i] -= d * c0[i];
    n = sqrtf(c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2]);
    if (n > 0) for (float &v : c1) v /= n;
    c2[0] = c0[1] * c1[2] - c0[2] * c1[1]; c2[1] = c0[2] * c1[0] - c0[0] * c1[2]; c2[2] = c0[0] * c1[1] - c0[1] * c1[0];
    for (int i = 0; i < 3; i++) { M.m[i] = c0[i]; M.m[4 + i] = c1[i]; M.m[8 + i] = c2[i]; }
    M.m[3] = M.m[7] = M.m[11] = 0.0f; M.m[15] = 1.0f;
    paramsFromM();
  }
};

class ITMSceneParams {
 public:
  float voxelSize, viewFrustum_min, viewFrustum_max, mu;
  int maxW;
  bool stopIntegratingAtMaxW;
  ITMSceneParams(float mu_, int maxW_, float voxelSize_, float vfMin, float vfMax, bool stopAtMax)
      : voxelSize(voxelSize_), viewFrustum_min(vfMin), viewFrustum_max(vfMax), mu(mu_), maxW(maxW_),
        stopIntegratingAtMaxW(stopAtMax) {}
};

/// ITMLibSettings is default-constructed by the reference and never edited (SystemEntry.cpp:238-243), so whatever
/// scene parameters a run uses are the ones this constructor sets.  The fork's defaults are not knowable; these are
/// the upstream InfiniTAM v2 defaults (5 mm voxels, 3 m frustum: an indoor RGB-D set-up), with the device fixed to HIP.
/// A KITTI-scale run needs other values: either two lines after SystemEntry.cpp:238
/// (`driver_settings->sceneParams = ITMSceneParams(mu, maxW, voxelSize, vfMin, vfMax, false);`), or -- with the
/// reference sources untouched -- the environment of the process, read here once per construction:
///   DSLAM_VOXEL_SIZE, DSLAM_MU, DSLAM_FRUSTUM_MIN, DSLAM_FRUSTUM_MAX (metres), DSLAM_MAX_W, DSLAM_USE_SWAPPING,
///   DSLAM_USE_BILATERAL_FILTER, DSLAM_LOCAL_BLOCKS, DSLAM_BUCKETS, DSLAM_EXCESS, DSLAM_DEVICE
/// (unset or unparsable variables leave the default).
class ITMLibSettings {
  static void envFloat(const char *name, float &v) {
    const char *s = getenv(name);
    if (s == nullptr || *s == 0) return;
    char *end = nullptr;
    const float x = strtof(s, &end);
    if (end != s && x > 0.0f) v = x;
  }
  static void envInt(const char *name, int &v) {
    const char *s = getenv(name);
    if (s == nullptr || *s == 0) return;
    char *end = nullptr;
    const long x = strtol(s, &end, 0);
    if (end != s && x >= 0) v = (int)x;
  }
  static void envBool(const char *name, bool &v) { int x = v ? 1 : 0; envInt(name, x); v = x != 0; }

 public:
  typedef enum { DEVICE_CPU, DEVICE_CUDA, DEVICE_HIP } DeviceType;
  DeviceType deviceType;
  bool useSwapping, useApproximateRaycast, useBilateralFilter, modelSensorNoise, skipPoints;
  ITMSceneParams sceneParams;
  // depth tracker (upstream ITMLibSettings defaults)
  typedef enum { TRACKER_ITERATION_ROTATION = 1, TRACKER_ITERATION_TRANSLATION = 2, TRACKER_ITERATION_BOTH = 3, TRACKER_ITERATION_NONE = 4 } TrackerIterationType;
  int noHierarchyLevels, noICPRunTillLevel;
  TrackerIterationType trackingRegime[8];
  float depthTrackerICPThreshold, depthTrackerTerminationThreshold;
  int hipDeviceIndex;
  bool meshWithColour;  ///< SaveCurrSceneToMesh writes `v x y z r g b` lines (see ITMMesh)
  // pool sizes (ITMLibDefines.h constants upstream; runtime here)
  int numLocalBlocks, numBuckets, numExcess;
  ITMLibSettings()
      : deviceType(DEVICE_HIP), useSwapping(false), useApproximateRaycast(false), useBilateralFilter(false),
        modelSensorNoise(false), skipPoints(true), sceneParams(0.02f, 100, 0.005f, 0.2f, 3.0f, false),
        noHierarchyLevels(5), noICPRunTillLevel(0), depthTrackerICPThreshold(0.1f * 0.1f),
        depthTrackerTerminationThreshold(1e-3f), hipDeviceIndex(0), meshWithColour(true),
        numLocalBlocks(SDF_LOCAL_BLOCK_NUM), numBuckets(SDF_BUCKET_NUM), numExcess(SDF_EXCESS_LIST_SIZE) {
    trackingRegime[0] = TRACKER_ITERATION_BOTH; trackingRegime[1] = TRACKER_ITERATION_BOTH;
    trackingRegime[2] = TRACKER_ITERATION_ROTATION; trackingRegime[3] = TRACKER_ITERATION_ROTATION;
    trackingRegime[4] = TRACKER_ITERATION_ROTATION;
    for (int i = 5; i < 8; i++) trackingRegime[i] = TRACKER_ITERATION_NONE;
    envFloat("DSLAM_VOXEL_SIZE", sceneParams.voxelSize); envFloat("DSLAM_MU", sceneParams.mu);
    envFloat("DSLAM_FRUSTUM_MIN", sceneParams.viewFrustum_min); envFloat("DSLAM_FRUSTUM_MAX", sceneParams.viewFrustum_max);
    envInt("DSLAM_MAX_W", sceneParams.maxW);
    envBool("DSLAM_USE_SWAPPING", useSwapping); envBool("DSLAM_USE_BILATERAL_FILTER", useBilateralFilter);
    envInt("DSLAM_LOCAL_BLOCKS", numLocalBlocks); envInt("DSLAM_BUCKETS", numBuckets); envInt("DSLAM_EXCESS", numExcess);
    envInt("DSLAM_DEVICE", hipDeviceIndex);
  }
};

class ITMView {
 public:
  const ITMRGBDCalib *calib;
  ITMUChar4Image *rgb;   ///< host mirror of the colour image (InfiniTamDriver.h:217)
  ITMFloatImage *depth;  ///< host mirror of the metric depth (InfiniTamDriver.h:218)
  dslam_view *handle;
  double timestamp;
  ITMView(const ITMRGBDCalib *c, Vector2i sz_rgb, Vector2i sz_d, dslam_engine *eng) : calib(c), handle(nullptr), timestamp(0) {
    rgb = new ITMUChar4Image(sz_rgb, true, true);
    depth = new ITMFloatImage(sz_d, true, true);
    dslam_check(dslam_view_create(eng, sz_rgb.x, sz_rgb.y, sz_d.x, sz_d.y, &handle), "dslam_view_create");
    // the device images are the master copies; the host mirrors are filled on the first GetData after an update
    dslam_view *h = handle;
    rgb->SetHostPull([eng, h](Vector4u *dst) { dslam_check(dslam_download_view_rgba(eng, h, &dst->x), "dslam_download_view_rgba"); });
    depth->SetHostPull([eng, h](float *dst) { dslam_check(dslam_download_view_depth(eng, h, dst), "dslam_download_view_depth"); });
  }
  void MarkHostStale() { rgb->MarkHostStale(); depth->MarkHostStale(); }
  ~ITMView() { dslam_view_destroy(handle); delete rgb; delete depth; }
};

/// The pool and list counters the reference reads as plain `int` members -- scene->localVBA.lastFreeBlockId
/// (InfiniTamDriver.h:345), renderState_vh->noVisibleEntries (:210), denseMapper->GetDecayedBlockCount() (:368) -- live on
/// the device (kernels chain on them without a host round trip).  The reference reads them in three places only, none of
/// them between UpdateView / IntegrateLocalMap / SlideWindow* / Decay (DenseSlam.cpp:210-232), so the mirror does not
/// fetch them after every call: a read fetches them (dslam_get_stats: one small copy + a wait for the engine's stream)
/// if a call has been enqueued since the last fetch.  Same pattern as MemoryBlock::SetHostPull / MarkHostStale.
/// A read is also a synchronising call in the sense of dslam_fusion.h: it throws what the device reported late.
class DeviceCounters {
  dslam_engine *eng_;
  const dslam_scene *scene_;
  const dslam_render_state *rs_;
  mutable bool stale_;
  mutable dslam_stats st_;
 public:
  DeviceCounters(dslam_engine *e, const dslam_scene *s, const dslam_render_state *r) : eng_(e), scene_(s), rs_(r), stale_(true) { memset(&st_, 0, sizeof(st_)); }
  void MarkStale() { stale_ = true; }
  const dslam_stats &Get() const {
    if (stale_) {
      dslam_check(dslam_get_stats(eng_, scene_, rs_, &st_), "dslam_get_stats");
      stale_ = false;
    }
    return st_;
  }
};
/// an `int` member of the reference's objects whose value is one field of DeviceCounters (reads as int; never assigned by
/// the reference's callers, so there is no assignment)
class LazyCounter {
  const DeviceCounters *src_;
  int32_t dslam_stats::*field_;
 public:
  LazyCounter() : src_(nullptr), field_(nullptr) {}
  void Bind(const DeviceCounters *src, int32_t dslam_stats::*field) { src_ = src; field_ = field; }
  operator int() const { return src_->Get().*field_; }
 private:
  LazyCounter(const LazyCounter &);
  LazyCounter &operator=(const LazyCounter &);
};

class ITMRenderState;
class ITMTrackingState {
 public:
  ITMPose *pose_d;
  ITMPose *pose_pointCloud;                ///< the pose the ICP maps were rendered from (set by Prepare)
  ITMFloat4Image *pointsMap, *normalsMap;  ///< ICP maps filled by trackingController->Prepare (host copies)
  ITMRenderState *preparedWith;            ///< render state whose device-resident ICP maps Track reads
  int age_pointCloud;
  explicit ITMTrackingState(Vector2i sz) : preparedWith(nullptr), age_pointCloud(-1) {
    pose_d = new ITMPose();
    pose_pointCloud = new ITMPose();
    pointsMap = new ITMFloat4Image(sz, true, true);
    normalsMap = new ITMFloat4Image(sz, true, true);
  }
  ~ITMTrackingState() { delete pose_d; delete pose_pointCloud; delete pointsMap; delete normalsMap; }
};

class ITMRenderState {
 public:
  dslam_render_state *handle;
  ITMUChar4Image *raycastImage;
  virtual ~ITMRenderState() { dslam_render_state_destroy(handle); delete raycastImage; }
  /// a call that can change this render state's visible list has been enqueued
  virtual void MarkCountersStale() {}
 protected:
  ITMRenderState() : handle(nullptr), raycastImage(nullptr) {}
};

class ITMRenderState_VH : public ITMRenderState {
  DeviceCounters counters_;
 public:
  LazyCounter noVisibleEntries;  ///< InfiniTamDriver.h:209-210 (reads as int; fetched from the device when stale)
  ITMRenderState_VH(dslam_engine *eng, const dslam_scene *scene, Vector2i sz) : counters_(eng, scene, nullptr) {
    dslam_check(dslam_render_state_create(eng, scene, sz.x, sz.y, &handle), "dslam_render_state_create");
    counters_ = DeviceCounters(eng, scene, handle);
    noVisibleEntries.Bind(&counters_, &dslam_stats::no_visible_entries);
    raycastImage = new ITMUChar4Image(sz, true, true);
  }
  void MarkCountersStale() { counters_.MarkStale(); }
};

class ITMLocalVBA {
 public:
  LazyCounter lastFreeBlockId;  ///< InfiniTamDriver.h:345 (reads as int)
  int allocatedSize;
};

class ITMVoxelBlockHash {
  int numBlocks_;
 public:
  LazyCounter lastFreeExcessListId;
  explicit ITMVoxelBlockHash(int n) : numBlocks_(n) {}
  int getNumAllocatedVoxelBlocks() const { return numBlocks_; }  ///< InfiniTamDriver.h:345,350
};

template <class TVoxel, class TIndex> class ITMScene {
  DeviceCounters *counters_;
 public:
  const ITMSceneParams *sceneParams;
  TIndex index;
  ITMLocalVBA localVBA;
  dslam_scene *handle;
  ITMScene(const ITMLibSettings *settings, dslam_engine *eng) : counters_(nullptr), sceneParams(&settings->sceneParams), index(settings->numLocalBlocks), handle(nullptr) {
    dslam_scene_params p;
    memset(&p, 0, sizeof(p));
    p.voxel_size = sceneParams->voxelSize; p.mu = sceneParams->mu; p.max_w = sceneParams->maxW;
    p.frustum_min = sceneParams->viewFrustum_min; p.frustum_max = sceneParams->viewFrustum_max;
    p.stop_integrating_at_max_w = sceneParams->stopIntegratingAtMaxW;
    p.num_local_blocks = settings->numLocalBlocks; p.num_buckets = settings->numBuckets; p.num_excess = settings->numExcess;
    p.use_swapping = settings->useSwapping;
    dslam_check(dslam_scene_create(eng, &p, nullptr, &handle), "dslam_scene_create");
    localVBA.allocatedSize = settings->numLocalBlocks;
    counters_ = new DeviceCounters(eng, handle, nullptr);
    localVBA.lastFreeBlockId.Bind(counters_, &dslam_stats::last_free_block_id);
    index.lastFreeExcessListId.Bind(counters_, &dslam_stats::last_free_excess_id);
  }
  ~ITMScene() { dslam_scene_destroy(handle); delete counters_; }
  /// a call that can change the pool counters the driver reads (InfiniTamDriver.h:344-351, 366-370) has been enqueued:
  /// the next read of one of them fetches all of them
  void MarkCountersStale(ITMRenderState *rs = nullptr) {
    counters_->MarkStale();
    if (rs) rs->MarkCountersStale();
  }
  const dslam_stats &Counters() const { return counters_->Get(); }
};

/// ITMMesh: the triangle list SaveCurrSceneToMesh writes (DenseSlam.cpp:638-643).  Host-side copy of the mesh the
/// engine built on the device; WriteOBJ / WriteSTL keep upstream's file layout (three `v` lines per triangle, faces
/// listed with the vertex order reversed).  With colours each `v` line carries r g b in [0, 1] after the position,
/// the coloured-OBJ convention of the DynSLAM lineage this fork descends from (not knowable from the reference
/// tree; ITMLibSettings::meshWithColour = false gives upstream v2's plain lines).
class ITMMesh {
 public:
  struct Triangle { Vector3f p0, p1, p2; };
  unsigned noTotalTriangles;
  unsigned noMaxTriangles;
  std::vector<Triangle> triangles, colours;
  explicit ITMMesh(unsigned maxTriangles) : noTotalTriangles(0), noMaxTriangles(maxTriangles) {}
  void WriteOBJ(const char *fileName) const {
    FILE *f = fopen(fileName, "w+");
    if (f == nullptr) throw std::runtime_error(std::string("ITMMesh::WriteOBJ: cannot open ") + fileName);
    const bool col = colours.size() == triangles.size() && !colours.empty();
    for (unsigned i = 0; i < noTotalTriangles; i++) {
      const Vector3f *p = &triangles[i].p0;
      for (int k = 0; k < 3; k++) {
        if (col) {
          const Vector3f &c = (&colours[i].p0)[k];
          fprintf(f, "v %f %f %f %f %f %f\n", p[k].x, p[k].y, p[k].z, c.x, c.y, c.z);
        } else {
          fprintf(f, "v %f %f %f\n", p[k].x, p[k].y, p[k].z);
        }
      }
    }
    for (unsigned i = 0; i < noTotalTriangles; i++) fprintf(f, "f %u %u %u\n", i * 3 + 2 + 1, i * 3 + 1 + 1, i * 3 + 0 + 1);
    fclose(f);
  }
  void WriteSTL(const char *fileName) const {
    FILE *f = fopen(fileName, "wb+");
    if (f == nullptr) throw std::runtime_error(std::string("ITMMesh::WriteSTL: cannot open ") + fileName);
    char header[80];
    memset(header, ' ', sizeof(header));
    fwrite(header, 1, sizeof(header), f);
    fwrite(&noTotalTriangles, sizeof(unsigned), 1, f);
    const float zero[3] = {0.0f, 0.0f, 0.0f};
    const short attribute = 0;
    for (unsigned i = 0; i < noTotalTriangles; i++) {
      fwrite(zero, sizeof(float), 3, f);  // facet normal left to the reader, as upstream does
      fwrite(&triangles[i].p2, sizeof(float), 3, f);
      fwrite(&triangles[i].p1, sizeof(float), 3, f);
      fwrite(&triangles[i].p0, sizeof(float), 3, f);
      fwrite(&attribute, sizeof(short), 1, f);
    }
    fclose(f);
  }
};

}  // namespace Objects

namespace Engine {
using namespace Objects;
/// ITMLocalMap {scene, renderState, trackingState, estimatedGlobalPose} (InfiniTamDriver.h:153,174,191,198,209,252-254)
class ITMLocalMap {
 public:
  ITMScene<ITMVoxel, ITMVoxelIndex> *scene;
  ITMRenderState *renderState;
  ITMTrackingState *trackingState;
  ITMPose estimatedGlobalPose;
  ITMLocalMap(const ITMLibSettings *settings, dslam_engine *eng, Vector2i trackedImageSize) {
    scene = new ITMScene<ITMVoxel, ITMVoxelIndex>(settings, eng);
    renderState = new ITMRenderState_VH(eng, scene->handle, trackedImageSize);
    trackingState = new ITMTrackingState(trackedImageSize);
  }
  ~ITMLocalMap() { delete renderState; delete trackingState; delete scene; }
};
}  // namespace Engine
}  // namespace ITMLib
