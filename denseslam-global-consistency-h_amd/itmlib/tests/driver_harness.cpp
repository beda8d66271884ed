// driver_harness.cpp -- stands in for the reference's InfiniTamDriver / DenseSlam host code to prove the ITMLib shim is
// a working drop-in: it derives from ITMLib::Engine::ITMMainEngine exactly like InfiniTamDriver
// (InfiniTamDriver.h:90-116) and issues the same member calls with the same expressions (cited per method), minus
// OpenCV / Eigen / Pangolin, which are not in this image.
//
//   driver_harness <frames.bin> <out.bin> <decay:0|1> <slide_max_age|-1> [keyframes.bin]
// keyframes.bin (optional, switches DenseSlam::OnlineCorrection on; frames.bin poses are then used as Twc = M_d^-1):
//             int32 CorrectionNum, StartToCorrectionNum, then per frame: int32 n, n x { double timestamp,
//             float Twc[16] (column-major), int32 isBad } = ORB-SLAM2's keyframes at that moment
// frames.bin: int32 W, H, N, then per frame: uint8 rgba[W*H*4], int16 depth_mm[W*H], float M_d[16] (column-major),
//             then float intr[4], then scene params: float voxel, mu, fmin, fmax, int32 maxW, nLocal, nBuckets, nExcess
// out.bin:    int32 lastFreeBlockId, noVisibleEntries, usedBytesLo, decayedBlocks; uint64 fnv(hash table), fnv(voxels);
//             float depth[W*H]; uint8 colour[W*H*4]; float trackedM[16] (TrackLocalMap of the last frame, started
//             from the pose of the frame before it);
//             with keyframes.bin, per frame: int32 nCorrected, double ts[nCorrected] (in re-fusion order),
//             int32 nCulled, int32 databaseSize;
//             trailer: uint64 fnv1a(GetImage(kRaycastImage = InfiniTAM_IMAGE_SCENERAYCAST) after Prepare), int32 its
//             non-zero bytes before the first Prepare (must be 0), int32 after;
//             uint64 fnv1a(view->rgb host mirror), uint64 fnv1a(view->depth host mirror),
//             int32 valid points, int32 valid normals of the tracking state's host ICP maps
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <vector>
#include <sched.h>

#include "DenseSLAM/OnlineCorrection.h"
#include "ITMLib/Engine/ITMMainEngine.h"

using namespace ITMLib::Engine;
using namespace ITMLib::Objects;

struct VoxelDecayParams { bool enabled; int min_decay_age, max_decay_weight; };
struct SlideWindowParams { bool enabled; int max_age; };
using SparsetoDense::OnlineCorrectionParams;

class DriverHarness : public ITMMainEngine {
 public:
  DriverHarness(const ITMLibSettings *settings, const ITMRGBDCalib *calib, const Vector2i &sz, VoxelDecayParams d, SlideWindowParams s,
                OnlineCorrectionParams oc = OnlineCorrectionParams(false, 0, 0))
      : ITMMainEngine(settings, calib, sz, sz), rgb_itm_(new ITMUChar4Image(sz, true, true)),
        raw_depth_itm_(new ITMShortImage(sz, true, true)), voxel_decay_params_(d), slide_window_params_(s),
        online_correction_params_(oc) {}
  ~DriverHarness() { delete rgb_itm_; delete raw_depth_itm_; }

  // InfiniTamDriver::UpdateView (InfiniTamDriver.cpp:280-288): CvToItm x 2 (here: plain copies into the driver's own images,
  // the cheapest stand-in for the reference's per-pixel loops), then viewBuilder->UpdateView
  void UpdateView(const uint8_t *rgba, const int16_t *depth, double timestamp, double *fill_us = nullptr) {
    const auto t0 = std::chrono::steady_clock::now();
    memcpy(rgb_itm_->GetData(MEMORYDEVICE_CPU), rgba, rgb_itm_->dataSize * 4);
    memcpy(raw_depth_itm_->GetData(MEMORYDEVICE_CPU), depth, raw_depth_itm_->dataSize * 2);
    if (fill_us) *fill_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    this->viewBuilder->UpdateView(&view, rgb_itm_, raw_depth_itm_, timestamp, settings->useBilateralFilter);
  }
  // the same call for a keyframe kept in the device-resident store (DenseSlam.cpp:392,421 without the upload)
  void UpdateViewFromStore(const dslam_frame_store *store, int slot, double timestamp) {
    this->viewBuilder->UpdateViewFromStore(&view, store, slot, timestamp, settings->useBilateralFilter);
  }
  // InfiniTamDriver::SetPoseLocalMap (InfiniTamDriver.h:173-178); last_egomotion_ feeds only the GUI
  void SetPoseLocalMap(const ITMLocalMap *m, const Matrix4f &new_pose) {
    const Matrix4f Tcurrmap_w = m->estimatedGlobalPose.GetM();
    m->trackingState->pose_d->SetInvM(Tcurrmap_w * new_pose);
  }
  // InfiniTamDriver::SlideWindowDefusionPart (InfiniTamDriver.h:302-310)
  void SlideWindowDefusionPart(const ITMLocalMap *m) {
    if (slide_window_params_.enabled) {
      int maxSize = (slide_window_params_.max_age - online_correction_params_.StartToCorrectionNum) * online_correction_params_.CorrectionNum;
      denseMapper->SlideWindowDefusionPart(m->scene, m->renderState, slide_window_params_.max_age, maxSize);
    }
  }
  ITMVoxelMapGraphManager *GetMapManager() const { return this->mapManager; }  // InfiniTamDriver.h:136
  // InfiniTamDriver::IntegrateLocalMap (InfiniTamDriver.h:187-192)
  void IntegrateLocalMap(const ITMLocalMap *m, bool onlyUpdateVisibleList = false, bool isDefusion = false) const {
    this->denseMapper->SetFusionWeightParams(fusion_weight_params_);
    this->denseMapper->ProcessFrame(this->view, m->trackingState, m->scene, m->renderState, onlyUpdateVisibleList, isDefusion);
  }
  // InfiniTamDriver::DeIntegrateLocalMap (InfiniTamDriver.h:194-199)
  void DeIntegrateLocalMap(const ITMLocalMap *m) const {
    this->denseMapper->SetFusionWeightParams(fusion_weight_params_);
    this->denseMapper->DeProcessFrame(this->view, m->trackingState, m->scene, m->renderState);
  }
  // (extension) the visible list of the fusion that has just run, kept with the keyframe; the re-fusion loop as one batch
  void KeepVisibleList(const ITMLocalMap *m, dslam_frame_store *store, int slot) const {
    this->denseMapper->KeepVisibleList(store, slot, m->scene, m->renderState);
  }
  void ReIntegrateLocalMapBatch(const ITMLocalMap *m, dslam_frame_store *store, int n, const int *slots, const Matrix4f *oldTwc,
                                const Matrix4f *newTwc, const double *) const {
    const Matrix4f Tcurrmap_w = m->estimatedGlobalPose.GetM();
    std::vector<Matrix4f> oldM(n), newM(n);
    ITMPose tmp;
    for (int k = 0; k < n; k++) {   // the matrices SetPoseLocalMap would have put into trackingState->pose_d
      tmp.SetInvM(Tcurrmap_w * oldTwc[k]); oldM[k] = tmp.GetM();
      tmp.SetInvM(Tcurrmap_w * newTwc[k]); newM[k] = tmp.GetM();
    }
    this->denseMapper->SetFusionWeightParams(fusion_weight_params_);
    this->denseMapper->ReProcessFrames(this->view, m->scene, m->renderState, store, n, slots, oldM.data(), newM.data());
    if (n > 0) m->trackingState->pose_d->SetInvM(Tcurrmap_w * newTwc[n - 1]);   // where the loop's last SetPoseLocalMap leaves it
  }
  // InfiniTamDriver::Decay (InfiniTamDriver.h:274-282)
  void Decay(const ITMLocalMap *m) {
    if (voxel_decay_params_.enabled)
      denseMapper->Decay(m->scene, m->renderState, voxel_decay_params_.max_decay_weight, voxel_decay_params_.min_decay_age, true);
  }
  // InfiniTamDriver::SlideWindow (InfiniTamDriver.h:294-300)
  void SlideWindow(const ITMLocalMap *m) {
    if (slide_window_params_.enabled) denseMapper->SlideWindow(m->scene, m->renderState, slide_window_params_.max_age);
  }
  // InfiniTamDriver::PrepareNextStepLocalMap (InfiniTamDriver.h:208-220)
  void PrepareNextStepLocalMap(const ITMLocalMap *m) {
    const ITMRenderState_VH *rs = (ITMRenderState_VH *)(m->renderState);
    if (rs->noVisibleEntries > 0) this->trackingController->Prepare(m->trackingState, m->scene, this->view, m->renderState);
  }
  // InfiniTamDriver::TrackLocalMap (InfiniTamDriver.h:151-163); last_egomotion_ feeds only the GUI
  void TrackLocalMap(ITMLocalMap *m) { this->trackingController->Track(m->trackingState, this->view); }
  // InfiniTamDriver::GetLocalMapUsedMemoryBytes (InfiniTamDriver.h:344-347)
  size_t GetLocalMapUsedMemoryBytes(ITMLocalMap *m) {
    int num_used_blocks = m->scene->index.getNumAllocatedVoxelBlocks() - m->scene->localVBA.lastFreeBlockId;
    return sizeof(ITMVoxel) * SDF_BLOCK_SIZE3 * num_used_blocks;
  }
  size_t GetSavedDecayMemoryBytes() const { return denseMapper->GetDecayedBlockCount() * sizeof(ITMVoxel) * SDF_BLOCK_SIZE3; }
  // InfiniTamDriver::GetImage / GetFloatImage (InfiniTamDriver.cpp:229-277) with the pose given directly
  void GetImage(ITMUChar4Image *out, ITMMainEngine::GetImageType t, ITMPose &pose, const ITMLocalMap *m) {
    if (nullptr != this->view) {
      ITMIntrinsics intrinsics = this->viewBuilder->GetCalib()->intrinsics_d;
      ITMMainEngine::GetImage(out, nullptr, t, &pose, &intrinsics, m);
    }
  }
  void GetFloatImage(ITMFloatImage *out, ITMPose &pose, const ITMLocalMap *m) {
    if (nullptr != this->view) {
      ITMIntrinsics intrinsics = this->viewBuilder->GetCalib()->intrinsics_d;
      ITMMainEngine::GetImage(nullptr, out, ITMMainEngine::InfiniTAM_IMAGE_FREECAMERA_DEPTH, &pose, &intrinsics, m);
    }
  }
  const ITMView *GetView() const { return view; }

 private:
  ITMUChar4Image *rgb_itm_;
  ITMShortImage *raw_depth_itm_;
  WeightParams fusion_weight_params_;
  VoxelDecayParams voxel_decay_params_;
  SlideWindowParams slide_window_params_;
  OnlineCorrectionParams online_correction_params_;
};

static uint64_t fnv1a(const void *p, size_t n, uint64_t h = 1469598103934665603ull) {
  const uint8_t *b = (const uint8_t *)p;
  for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}

// The harness plays the reference's main thread, which fills the driver's page-locked images every frame: run where that memory
// is local (dslam_device_numa_node; DRIVER_HARNESS_NO_BIND=1 leaves the affinity alone).  On a two-socket box the 1.8 MB fill takes
// ~60 us on the GPU's node and ~160 us from the other socket.
static void bind_near_device(int device) {
  if (getenv("DRIVER_HARNESS_NO_BIND")) return;
  int node = -1;
  if (dslam_device_numa_node(device, &node) != DSLAM_OK || node < 0) return;
  char path[96];
  snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
  FILE *f = fopen(path, "r");
  if (!f) return;
  cpu_set_t now, want;
  CPU_ZERO(&want);
  if (sched_getaffinity(0, sizeof(now), &now) != 0) { fclose(f); return; }
  int a, b, picked = 0;
  while (fscanf(f, "%d", &a) == 1) {   // "0-63,128-191"
    b = a;
    int c = fgetc(f);
    if (c == '-') { if (fscanf(f, "%d", &b) != 1) break; c = fgetc(f); }
    for (int i = a; i <= b && i < CPU_SETSIZE; i++)
      if (CPU_ISSET(i, &now)) { CPU_SET(i, &want); picked++; }
    if (c != ',') break;
  }
  fclose(f);
  if (picked > 0) (void)sched_setaffinity(0, sizeof(want), &want);
}

int main(int argc, char **argv) {
  bind_near_device(getenv("DSLAM_DEVICE") ? atoi(getenv("DSLAM_DEVICE")) : 0);   // (before the frames are read: their pages should be local too)
  if (argc < 5) { fprintf(stderr, "usage: %s frames.bin out.bin decay slide_max_age\n", argv[0]); return 2; }
  FILE *f = fopen(argv[1], "rb");
  if (!f) { perror("frames"); return 2; }
  int32_t hdr[3];
  if (fread(hdr, 4, 3, f) != 3) return 2;
  const int W = hdr[0], H = hdr[1], N = hdr[2];
  std::vector<std::vector<uint8_t>> rgba(N, std::vector<uint8_t>((size_t)W * H * 4));
  std::vector<std::vector<int16_t>> depth(N, std::vector<int16_t>((size_t)W * H));
  std::vector<Matrix4f> poses(N);
  for (int i = 0; i < N; i++) {
    if (fread(rgba[i].data(), 1, rgba[i].size(), f) != rgba[i].size()) return 2;
    if (fread(depth[i].data(), 2, depth[i].size(), f) != depth[i].size()) return 2;
    if (fread(poses[i].m, 4, 16, f) != 16) return 2;
  }
  float intr[4], sp[4];
  int32_t ip[4];
  if (fread(intr, 4, 4, f) != 4 || fread(sp, 4, 4, f) != 4 || fread(ip, 4, 4, f) != 4) return 2;
  fclose(f);

  // optional ORB-SLAM2 keyframe snapshots, one set per fused frame
  OnlineCorrectionParams oc(false, 0, 0);
  std::vector<std::vector<SparsetoDense::MapKeyFrame>> keyframes(N);
  if (argc >= 6) {
    FILE *k = fopen(argv[5], "rb");
    if (!k) { perror("keyframes"); return 2; }
    int32_t cn[2];
    if (fread(cn, 4, 2, k) != 2) return 2;
    oc = OnlineCorrectionParams(true, cn[0], cn[1]);
    for (int i = 0; i < N; i++) {
      int32_t n;
      if (fread(&n, 4, 1, k) != 1) return 2;
      keyframes[i].resize(n);
      for (int j = 0; j < n; j++) {
        int32_t bad;
        if (fread(&keyframes[i][j].mTimeStamp, 8, 1, k) != 1 || fread(keyframes[i][j].poseInverse.m, 4, 16, k) != 16 || fread(&bad, 4, 1, k) != 1) return 2;
        keyframes[i][j].bad = bad != 0;
      }
    }
    fclose(k);
  }
  std::vector<int32_t> oc_counts;      // per frame: nCorrected, nCulled, database size
  std::vector<std::vector<double>> oc_order(N);

  try {
    ITMLibSettings *settings = new ITMLibSettings();  // SystemEntry.cpp:238
    if (getenv("DRIVER_HARNESS_SETTINGS_FROM_ENV") == nullptr) {
      settings->sceneParams = ITMSceneParams(sp[1], ip[0], sp[0], sp[2], sp[3], false);
      settings->numLocalBlocks = ip[1]; settings->numBuckets = ip[2]; settings->numExcess = ip[3];
    }  // else: exactly the reference's two lines -- the settings object is used as constructed (DSLAM_* environment)
    // CreateItmCalib (InfiniTamDriver.cpp:55-81)
    ITMRGBDCalib *calib = new ITMRGBDCalib;
    ITMIntrinsics intrinsics;
    intrinsics.SetFrom(intr[0], intr[1], intr[2], intr[3], (float)W, (float)H);
    calib->intrinsics_rgb = intrinsics; calib->intrinsics_d = intrinsics;
    Matrix4f identity; identity.setIdentity();
    calib->trafo_rgb_to_depth.SetFrom(identity);
    calib->disparityCalib.SetFrom(1.0f / 1000.0f, 0.0f, ITMDisparityCalib::TRAFO_AFFINE);

    VoxelDecayParams dp = {atoi(argv[3]) != 0, 2, 1};
    if (const char *d = getenv("DRIVER_HARNESS_DECAY")) sscanf(d, "%d,%d", &dp.max_decay_weight, &dp.min_decay_age);   // "maxWeight,minAge" 
    SlideWindowParams sw = {atoi(argv[4]) >= 0, atoi(argv[4])};
    DriverHarness drv(settings, calib, Vector2i(W, H), dp, sw, oc);

    // before the first keyframe: GetImage must be a no-op, not a crash
    ITMFloatImage out_float(Vector2i(W, H), true, true);
    ITMUChar4Image out_rgba(Vector2i(W, H), true, true);
    ITMPose free_pose;
    drv.GetFloatImage(&out_float, free_pose, nullptr);

    // DenseSlam::ProcessFrame, first keyframe (DenseSlam.cpp:133-152)
    int idx = drv.GetMapManager()->createNewLocalMap();
    ITMPose tempPose;
    drv.GetMapManager()->setEstimatedGlobalPose(idx, tempPose);
    ITMLocalMap *currentLocalMap = drv.GetMapManager()->getLocalMap(idx);

    if (oc.enabled) {
      // DenseSlam::ProcessFrame with online_correction: 1 (DenseSlam.cpp:156-158,178-181,210-232)
      struct Recorder {  // forwards to the driver and notes the order keyframes are re-fused in
        DriverHarness &d; std::vector<double> *order; double last_ts;
        void SetPoseLocalMap(const ITMLocalMap *m, const Matrix4f &p) { d.SetPoseLocalMap(m, p); }
        void UpdateViewFromStore(const dslam_frame_store *s, int slot, double ts) { last_ts = ts; d.UpdateViewFromStore(s, slot, ts); }
        void DeIntegrateLocalMap(const ITMLocalMap *m) { d.DeIntegrateLocalMap(m); }
        void IntegrateLocalMap(const ITMLocalMap *m, bool a, bool b) { order->push_back(last_ts); d.IntegrateLocalMap(m, a, b); }
        void ReIntegrateLocalMapBatch(const ITMLocalMap *m, dslam_frame_store *s, int n, const int *slots, const Matrix4f *o, const Matrix4f *nw,
                                      const double *ts) {
          for (int k = 0; k < n; k++) order->push_back(ts[k]);
          d.ReIntegrateLocalMapBatch(m, s, n, slots, o, nw, ts);
        }
      };
      const bool batched = getenv("DRIVER_HARNESS_BATCHED") != nullptr;   // OnlineCorrectionBatched instead of OnlineCorrection
      SparsetoDense::FusionFrameDataBase mfusionFrameDataBase(drv.GetDslamEngine(), Vector2i(W, H), Vector2i(W, H), N);
      if (batched) mfusionFrameDataBase.EnableVisibleLists(currentLocalMap->scene->handle);
      for (int i = 0; i < N; i++) {
        const double currBAKFTime = (double)i;
        Matrix4f orbSLAM2_Pose;
        poses[i].inv(orbSLAM2_Pose);                                            // Twc
        drv.UpdateView(rgba[i].data(), depth[i].data(), currBAKFTime);          // the one upload of this keyframe
        mfusionFrameDataBase.InsertFromView(currBAKFTime, orbSLAM2_Pose, drv.GetView());  // DenseSlam.cpp:156-157
        Recorder rec{drv, &oc_order[i], 0.0};
        int culled = 0;
        const int corrected = batched ? mfusionFrameDataBase.OnlineCorrectionBatched(rec, currentLocalMap, keyframes[i], oc, &culled)
                                      : mfusionFrameDataBase.OnlineCorrection(rec, currentLocalMap, keyframes[i], oc, &culled);  // :178-181
        drv.SetPoseLocalMap(currentLocalMap, orbSLAM2_Pose);                    // :189
        if (mfusionFrameDataBase.entries().count(currBAKFTime)) {
          drv.UpdateViewFromStore(mfusionFrameDataBase.store(), mfusionFrameDataBase.entries().at(currBAKFTime).slot, currBAKFTime);  // :212
          drv.IntegrateLocalMap(currentLocalMap);                               // :213
          if (batched) drv.KeepVisibleList(currentLocalMap, mfusionFrameDataBase.store(), mfusionFrameDataBase.SlotOf(currBAKFTime));
        }
        if ((int)mfusionFrameDataBase.size() > sw.max_age && sw.enabled) {      // :215-225
          drv.SlideWindow(currentLocalMap);
          for (int k = 0; k < oc.CorrectionNum; k++) drv.SlideWindowDefusionPart(currentLocalMap);
          mfusionFrameDataBase.SlideWindowPose(sw.max_age);
        }
        drv.Decay(currentLocalMap);                                             // :227-232
        oc_counts.push_back(corrected); oc_counts.push_back(culled); oc_counts.push_back((int32_t)mfusionFrameDataBase.size());
      }
    }
    int fused = 0;
    auto t_loop0 = std::chrono::steady_clock::now();
    const bool raycast_each = getenv("DRIVER_HARNESS_RAYCAST_EACH_FRAME") != nullptr;
    // timing starts at this keyframe (the ones before it warm the engine up and fill the window); the loop's last call is
    // followed by a counter read -- a synchronising call -- so that the time covers the work, not only its enqueueing
    const int time_from = getenv("DRIVER_HARNESS_TIME_FROM") ? atoi(getenv("DRIVER_HARNESS_TIME_FROM")) : 0;
    float image_probe = 0.0f;
    double phase_us[3] = {0, 0, 0};  // UpdateView, fusion + window + decay, per-keyframe raycast
    double fill_us = 0;              // ... of UpdateView: the copies into the driver's images (CvToItm's stand-in: the caller's own work)
    auto lap = [](std::chrono::steady_clock::time_point &t) {
      const auto now = std::chrono::steady_clock::now();
      const double us = std::chrono::duration<double, std::micro>(now - t).count();
      t = now;
      return us;
    };
    for (int i = 0; i < N && !oc.enabled; i++) {
      if (i == time_from && i > 0) {
        (void)drv.GetLocalMapUsedMemoryBytes(currentLocalMap);   // (counter read: waits for the keyframes before)
        phase_us[0] = phase_us[1] = phase_us[2] = fill_us = 0;
        t_loop0 = std::chrono::steady_clock::now();
      }
      auto t = std::chrono::steady_clock::now();
      currentLocalMap->trackingState->pose_d->SetM(poses[i]);                 // SetPoseLocalMap (InfiniTamDriver.h:173-178)
      drv.UpdateView(rgba[i].data(), depth[i].data(), (double)i, &fill_us);   // DenseSlam.cpp:212
      phase_us[0] += lap(t);
      drv.IntegrateLocalMap(currentLocalMap);                                 // DenseSlam.cpp:213
      fused++;
      if (sw.enabled && fused > sw.max_age) drv.SlideWindow(currentLocalMap);  // DenseSlam.cpp:215-225
      drv.Decay(currentLocalMap);                                             // DenseSlam.cpp:227-232
      phase_us[1] += lap(t);
      if (raycast_each) {                                                     // SaveRaycastDepth's per-keyframe raycast
        free_pose.SetM(poses[i]);
        drv.GetFloatImage(&out_float, free_pose, currentLocalMap);
        // the reference converts the image right away (ItmDepthToCv / FloatDepthmapToInt16, InfiniTamDriver.cpp:276,
        // DenseSlam.cpp:586-592): read it through the accessor the reference uses
        image_probe += out_float.GetData(MEMORYDEVICE_CPU)[((size_t)H / 2) * W + W / 2];
        phase_us[2] += lap(t);
      }
    }
    const size_t used_bytes_after_loop = N > 0 && !oc.enabled ? drv.GetLocalMapUsedMemoryBytes(currentLocalMap) : 0;   // (waits for the last keyframe)
    const double loop_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop0).count();
    if (!oc.enabled && N > 0) {
      const int nt = N - (time_from > 0 && time_from < N ? time_from : 0);
      printf("driver_harness loop: %d keyframes in %.3f ms (%.1f us per keyframe; host time inside the calls: UpdateView %.1f of which image fill %.1f, "
             "fusion + window + decay %.1f, raycast %.1f; %zu bytes in use, probe %.3f)\n",
             nt, loop_s * 1e3, loop_s * 1e6 / nt, phase_us[0] / nt, fill_us / nt, phase_us[1] / nt, phase_us[2] / nt, used_bytes_after_loop, image_probe);
    }
    free_pose.SetM(poses[N - 1]);
    drv.GetFloatImage(&out_float, free_pose, currentLocalMap);                // DenseSlam.h:146-153
    drv.GetImage(&out_rgba, ITMMainEngine::InfiniTAM_IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, free_pose, currentLocalMap);
    // PreviewType::kRaycastImage -> InfiniTAM_IMAGE_SCENERAYCAST (InfiniTamDriver.cpp:28-29): the grey tracking raycast;
    // nothing has drawn it before the first Prepare (the reference's ORB-SLAM2 mode never does), afterwards it is there
    ITMUChar4Image out_ray(Vector2i(W, H), true, true);
    auto nonzero = [&](ITMUChar4Image &img) {
      int32_t n = 0;
      const uint8_t *b = &img.GetData(MEMORYDEVICE_CPU)->x;
      for (size_t i = 0; i < (size_t)W * H * 4; i++) n += b[i] != 0;
      return n;
    };
    drv.GetImage(&out_ray, ITMMainEngine::InfiniTAM_IMAGE_SCENERAYCAST, free_pose, currentLocalMap);
    const int32_t ray_nonzero_before = nonzero(out_ray);
    drv.PrepareNextStepLocalMap(currentLocalMap);
    drv.GetImage(&out_ray, ITMMainEngine::InfiniTAM_IMAGE_SCENERAYCAST, free_pose, currentLocalMap);
    const int32_t ray_nonzero_after = nonzero(out_ray);
    const uint64_t ray_sum = fnv1a(out_ray.GetData(MEMORYDEVICE_CPU), (size_t)W * H * 4);
    // DenseSlam::ProcessFrame without ORB-SLAM2 odometry (DenseSlam.cpp:200-206): Prepare, UpdateView, TrackLocalMap.
    // The view still holds the last frame; start from the previous frame's pose and let the tracker pull it over.
    Matrix4f trackedM = poses[N - 1];
    if (N >= 2) {
      if (oc.enabled) drv.UpdateView(rgba[N - 1].data(), depth[N - 1].data(), (double)(N - 1));
      currentLocalMap->trackingState->pose_d->SetM(poses[N - 2]);
      drv.TrackLocalMap(currentLocalMap);
      trackedM = currentLocalMap->trackingState->pose_d->GetM();
    }

    dslam_engine *e = drv.GetDslamEngine();
    std::vector<dslam_hash_entry> hash((size_t)ip[2] + ip[3]);
    std::vector<dslam_voxel> vox((size_t)ip[1] * 512);
    ITMLib::dslam_check(dslam_download_hash_table(e, currentLocalMap->scene->handle, hash.data()), "download hash");
    ITMLib::dslam_check(dslam_download_voxel_blocks(e, currentLocalMap->scene->handle, 0, ip[1], vox.data()), "download voxels");

    FILE *o = fopen(argv[2], "wb");
    if (!o) { perror("out"); return 2; }
    const ITMRenderState_VH *rs = (const ITMRenderState_VH *)currentLocalMap->renderState;
    int32_t st[4] = {currentLocalMap->scene->localVBA.lastFreeBlockId, rs->noVisibleEntries,
                     (int32_t)(drv.GetLocalMapUsedMemoryBytes(currentLocalMap) & 0x7fffffff),
                     (int32_t)(drv.GetSavedDecayMemoryBytes() / (sizeof(ITMVoxel) * SDF_BLOCK_SIZE3))};
    uint64_t sums[2] = {fnv1a(hash.data(), hash.size() * sizeof(dslam_hash_entry)), fnv1a(vox.data(), vox.size() * sizeof(dslam_voxel))};
    fwrite(st, 4, 4, o);
    fwrite(sums, 8, 2, o);
    fwrite(out_float.GetData(MEMORYDEVICE_CPU), 4, (size_t)W * H, o);
    fwrite(out_rgba.GetData(MEMORYDEVICE_CPU), 4, (size_t)W * H, o);
    fwrite(trackedM.m, 4, 16, o);
    for (int i = 0; i < N && oc.enabled; i++) {
      const int32_t n = (int32_t)oc_order[i].size();
      fwrite(&n, 4, 1, o);
      fwrite(oc_order[i].data(), 8, oc_order[i].size(), o);
      fwrite(&oc_counts[3 * i + 1], 4, 2, o);
    }
    // DenseSlam::SaveStaticMap (DenseSlam.cpp:638-643): the mesh of the map as an OBJ file
    if (const char *obj = getenv("DRIVER_HARNESS_MESH_OBJ")) drv.SaveCurrSceneToMesh(obj, currentLocalMap->scene);
    if (const char *stl = getenv("DRIVER_HARNESS_MESH_STL")) {  // upstream's other writer, same triangle list
      ITMLib::Objects::ITMMesh mesh((unsigned)ip[1] * 32u);
      drv.MeshScene(&mesh, currentLocalMap->scene);
      mesh.WriteSTL(stl);
    }
    // trailer: the lazily filled host mirrors (view->rgb / view->depth of the last UpdateView, and the tracking
    // state's points map from the last Prepare), read through the same GetData calls the reference driver makes
    {
      const ITMView *v = drv.GetView();
      const size_t npx = (size_t)W * H;
      uint64_t mirror[2] = {fnv1a(v->rgb->GetData(MEMORYDEVICE_CPU), npx * 4), fnv1a(v->depth->GetData(MEMORYDEVICE_CPU), npx * 4)};
      const Vector4f *pts = currentLocalMap->trackingState->pointsMap->GetData(MEMORYDEVICE_CPU);
      const Vector4f *nrm = currentLocalMap->trackingState->normalsMap->GetData(MEMORYDEVICE_CPU);
      int32_t valid[2] = {0, 0};
      for (size_t i = 0; i < npx; i++) { valid[0] += pts[i].w > 0.0f; valid[1] += nrm[i].w == 0.0f; }
      const int32_t ray_counts[2] = {ray_nonzero_before, ray_nonzero_after};
      fwrite(&ray_sum, 8, 1, o);
      fwrite(ray_counts, 4, 2, o);
      fwrite(mirror, 8, 2, o);
      fwrite(valid, 4, 2, o);
    }
    fclose(o);
    printf("driver_harness ok: %d frames, lastFreeBlockId %d, visible %d, decayed %d\n", N, st[0], st[1], st[3]);
    delete calib;
    delete settings;
  } catch (const std::exception &ex) {
    fprintf(stderr, "driver_harness failed: %s\n", ex.what());
    return 1;
  }
  return 0;
}
