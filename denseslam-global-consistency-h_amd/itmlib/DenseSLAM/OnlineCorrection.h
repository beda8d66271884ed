// OnlineCorrection.h -- the host side of the reference's global-consistency step, written against the MI355X engine
// (SURVEY.md 8f N2).  It mirrors, name for name, the pieces of DenseSlam that keep the fused keyframes and re-fuse
// them after ORB-SLAM2 has moved their poses:
//
//   mfusionFrameDataBase / fusionFrameInfo      [REF DenseSlam.h:46-60,431-433]   -> FusionFrameDataBase
//   insert of the current keyframe              [REF DenseSlam.cpp:156-158]       -> Insert / InsertFromView
//   DenseSlam::OnlineCorrection                 [REF DenseSlam.cpp:298-432]       -> OnlineCorrection
//   DenseSlam::SlideWindowPose                  [REF DenseSlam.cpp:284-296]       -> SlideWindowPose
//   DenseSlam::is_identity_matrix               [REF DenseSlam.cpp:268-282]       -> is_identity_matrix
//
// What is different by design: the RGB and depth images of every keyframe live in HBM (dslam_frame_store), not in
// cv::Mat on the host, so the UpdateView in front of each DeIntegrate / Integrate is a pointer switch instead of a
// 1.8 MB upload, and the reference's erase-while-iterating over the std::map (undefined behaviour at
// DenseSlam.cpp:424) is an ordinary safe erase.  ORB-SLAM2 itself is outside the path: its keyframes arrive as
// plain (timestamp, Twc, isBad) records.
#pragma once
#include <cmath>
#include <functional>
#include <map>
#include <vector>

#include "../ITMLib/Engine/ITMMainEngine.h"

namespace SparsetoDense {

struct OnlineCorrectionParams {  // [REF VoxelDecayParams.h:28-36]
  bool enabled;
  int CorrectionNum;
  int StartToCorrectionNum;
  OnlineCorrectionParams(bool enabled_, int CorrectionNum_, int StartToCorrectionNum_)
      : enabled(enabled_), CorrectionNum(CorrectionNum_), StartToCorrectionNum(StartToCorrectionNum_) {}
};

/// fusionFrameInfo with the two cv::Mat images replaced by a slot of the device-resident store
struct fusionFrameInfo {
  Matrix4f poseinfo;  ///< Twc the keyframe was fused with
  int slot;           ///< dslam_frame_store slot holding rgbinfo / depthinfo
  short flaginfo;     ///< 1 once ORB-SLAM2's map was seen to contain the keyframe
};

/// mapKeyframeInfo [REF DenseSlam.h:62-70]
struct mapKeyframeInfo {
  double timestampinfo;
  Matrix4f poseinfok;
};

/// what OnlineCorrection reads of an ORB_SLAM2::KeyFrame: mTimeStamp, GetPoseInverse(), isBad()
struct MapKeyFrame {
  double mTimeStamp;
  Matrix4f poseInverse;  ///< Twc after local / global bundle adjustment
  bool bad;
};

inline bool is_identity_matrix(const Matrix4f &m) {
  for (int row = 0; row < 4; row++)
    for (int col = 0; col < 4; col++) {
      if (m.at(row, row) != 1.0f) return false;  // at(col, row): column-major storage, row == col here
      if (row != col && m.at(col, row) != 0.0f) return false;
    }
  return true;
}

/// right pose difference preDiff = prePose^-1 * currPose, its se(3) log, and sqrt(trace(E W E^T)) with
/// W = diag(.5,.5,.5,1) [REF DenseSlam.cpp:322-357], which expands to sqrt(|rot|^2 + |trans|^2)
inline bool PoseError(const Matrix4f &prePose, const Matrix4f &currPose, float *rightError) {
  Matrix4f preInv;
  prePose.inv(preInv);
  const Matrix4f poseDiff = preInv * currPose;
  if (is_identity_matrix(poseDiff)) return false;
  ITMLib::Objects::ITMPose tempDiff;
  tempDiff.SetInvM(poseDiff);
  const float *se3 = tempDiff.GetParams();  // tx, ty, tz, rx, ry, rz
  float traceValue = 0.0f;
  for (int i = 3; i < 6; i++) traceValue += se3[i] * se3[i];
  for (int i = 0; i < 3; i++) traceValue += se3[i] * se3[i];
  *rightError = sqrtf(traceValue);
  return true;
}

class FusionFrameDataBase {
 public:
  typedef std::map<double, fusionFrameInfo> Map;

  FusionFrameDataBase(dslam_engine *e, Vector2i sizeRgb, Vector2i sizeDepth, int capacity) : eng_(e), store_(nullptr) {
    ITMLib::dslam_check(dslam_frame_store_create(e, sizeRgb.x, sizeRgb.y, sizeDepth.x, sizeDepth.y, capacity, &store_), "dslam_frame_store_create");
    for (int i = capacity - 1; i >= 0; i--) free_.push_back(i);
  }
  ~FusionFrameDataBase() { dslam_frame_store_destroy(store_); }
  FusionFrameDataBase(const FusionFrameDataBase &) = delete;
  FusionFrameDataBase &operator=(const FusionFrameDataBase &) = delete;

  size_t size() const { return db_.size(); }
  const Map &entries() const { return db_; }
  dslam_frame_store *store() const { return store_; }

  /// mfusionFrameDataBase[currBAKFTime] = fusionFrameInfo(pose, rgb, depth, 0, ...) with host images
  void Insert(double timestamp, const Matrix4f &pose, const Vector4u *rgba, const short *depth) {
    const int slot = slotFor(timestamp);
    ITMLib::dslam_check(dslam_frame_store_put(eng_, store_, slot, &rgba->x, depth), "dslam_frame_store_put");
    db_[timestamp] = fusionFrameInfo{pose, slot, 0};
  }
  /// same, taking the images from the view they were just uploaded to (device-to-device)
  void InsertFromView(double timestamp, const Matrix4f &pose, const ITMLib::Objects::ITMView *view) {
    const int slot = slotFor(timestamp);
    ITMLib::dslam_check(dslam_frame_store_put_view(eng_, store_, slot, view->handle), "dslam_frame_store_put_view");
    db_[timestamp] = fusionFrameInfo{pose, slot, 0};
  }

  /// (extension) room for one visible list per keyframe; the driver then calls KeepVisibleList after every fusion of a
  /// keyframe, and OnlineCorrectionBatched can replace OnlineCorrection
  void EnableVisibleLists(const dslam_scene *scene) {
    ITMLib::dslam_check(dslam_frame_store_enable_lists(eng_, store_, scene), "dslam_frame_store_enable_lists");
  }
  int SlotOf(double timestamp) const { return db_.at(timestamp).slot; }

  /// DenseSlam::SlideWindowPose: drop the oldest entries until max_age remain
  void SlideWindowPose(int max_age) {
    int cullSize = (int)db_.size() - max_age;
    while (cullSize-- > 0 && !db_.empty()) erase(db_.begin());
  }

  /// DenseSlam::OnlineCorrection.  Driver supplies SetPoseLocalMap(map, Twc), UpdateViewFromStore(store, slot,
  /// timestamp), DeIntegrateLocalMap(map) and IntegrateLocalMap(map, onlyUpdateVisibleList, isDefusion) -- the calls
  /// the reference makes on InfiniTamDriver.  Returns the number of keyframes re-fused; *culled (optional) the
  /// number of keyframes taken out of the map because ORB-SLAM2 no longer has them.
  template <class Driver, class LocalMap>
  int OnlineCorrection(Driver &static_scene_, const LocalMap *currentLocalMap, const std::vector<MapKeyFrame> &currAllKeyFrame,
                       const OnlineCorrectionParams &online_correction_, int *culled = nullptr) {
    // error -> keyframe, largest error first; equal errors overwrite each other, as in the reference's std::map
    std::map<float, mapKeyframeInfo, std::greater<float>> mapPoseError;
    for (size_t i = 0; i < currAllKeyFrame.size(); i++) {
      const MapKeyFrame &kf = currAllKeyFrame[i];
      if (kf.bad) continue;
      Map::iterator fusioniter = db_.find(kf.mTimeStamp);
      if (fusioniter == db_.end()) continue;
      fusioniter->second.flaginfo = 1;
      float rightError;
      if (!PoseError(fusioniter->second.poseinfo, kf.poseInverse, &rightError)) continue;
      mapPoseError[rightError] = mapKeyframeInfo{kf.mTimeStamp, kf.poseInverse};
    }

    int countNum = 0;
    if ((int)mapPoseError.size() > online_correction_.StartToCorrectionNum - 1) {
      for (auto errorIter = mapPoseError.begin(); errorIter != mapPoseError.end(); ++errorIter) {
        const double timestamp = errorIter->second.timestampinfo;
        Map::iterator defusioniter = db_.find(timestamp);
        if (defusioniter != db_.end()) {
          // Deintegrate at the pose the keyframe was fused with
          static_scene_.SetPoseLocalMap(currentLocalMap, defusioniter->second.poseinfo);
          static_scene_.UpdateViewFromStore(store_, defusioniter->second.slot, timestamp);
          static_scene_.DeIntegrateLocalMap(currentLocalMap);
          // Reintegrate at the optimised pose (onlyUpdateVisibleList = false, isDefusion = true)
          defusioniter->second.poseinfo = errorIter->second.poseinfok;
          static_scene_.SetPoseLocalMap(currentLocalMap, defusioniter->second.poseinfo);
          static_scene_.IntegrateLocalMap(currentLocalMap, false, true);
          countNum++;
        }
        if (countNum > online_correction_.CorrectionNum - 1) break;
      }
    }

    // keyframes ORB-SLAM2 has culled never take part in its optimisation: take them out of the map and the database
    int n_culled = 0;
    for (Map::iterator iter = db_.begin(); iter != db_.end();) {
      if (iter->second.flaginfo == 0) {
        static_scene_.SetPoseLocalMap(currentLocalMap, iter->second.poseinfo);
        static_scene_.UpdateViewFromStore(store_, iter->second.slot, 0.0);
        static_scene_.DeIntegrateLocalMap(currentLocalMap);
        iter = erase(iter);
        n_culled++;
      } else {
        ++iter;
      }
    }
    if (culled) *culled = n_culled;
    return countNum;
  }

  /// (extension) OnlineCorrection with the re-fusions of the selected keyframes as ONE batch on the device.  Same selection,
  /// same order, same culling; the Driver supplies, instead of the per-keyframe calls,
  ///   ReIntegrateLocalMapBatch(map, store, n, slots, oldTwc, newTwc, timestamps)
  /// and must have kept every keyframe's visible list (EnableVisibleLists + KeepVisibleList after each fusion).
  template <class Driver, class LocalMap>
  int OnlineCorrectionBatched(Driver &static_scene_, const LocalMap *currentLocalMap, const std::vector<MapKeyFrame> &currAllKeyFrame,
                              const OnlineCorrectionParams &online_correction_, int *culled = nullptr) {
    std::map<float, mapKeyframeInfo, std::greater<float>> mapPoseError;
    for (size_t i = 0; i < currAllKeyFrame.size(); i++) {
      const MapKeyFrame &kf = currAllKeyFrame[i];
      if (kf.bad) continue;
      Map::iterator fusioniter = db_.find(kf.mTimeStamp);
      if (fusioniter == db_.end()) continue;
      fusioniter->second.flaginfo = 1;
      float rightError;
      if (!PoseError(fusioniter->second.poseinfo, kf.poseInverse, &rightError)) continue;
      mapPoseError[rightError] = mapKeyframeInfo{kf.mTimeStamp, kf.poseInverse};
    }
    std::vector<int> slots;
    std::vector<Matrix4f> oldTwc, newTwc;
    std::vector<double> stamps;
    if ((int)mapPoseError.size() > online_correction_.StartToCorrectionNum - 1) {
      for (auto errorIter = mapPoseError.begin(); errorIter != mapPoseError.end(); ++errorIter) {
        Map::iterator defusioniter = db_.find(errorIter->second.timestampinfo);
        if (defusioniter != db_.end()) {
          slots.push_back(defusioniter->second.slot);
          oldTwc.push_back(defusioniter->second.poseinfo);
          newTwc.push_back(errorIter->second.poseinfok);
          stamps.push_back(errorIter->second.timestampinfo);
          defusioniter->second.poseinfo = errorIter->second.poseinfok;
        }
        if ((int)slots.size() > online_correction_.CorrectionNum - 1) break;
      }
    }
    if (!slots.empty())
      static_scene_.ReIntegrateLocalMapBatch(currentLocalMap, store_, (int)slots.size(), slots.data(), oldTwc.data(), newTwc.data(), stamps.data());
    int n_culled = 0;
    for (Map::iterator iter = db_.begin(); iter != db_.end();) {
      if (iter->second.flaginfo == 0) {
        static_scene_.SetPoseLocalMap(currentLocalMap, iter->second.poseinfo);
        static_scene_.UpdateViewFromStore(store_, iter->second.slot, 0.0);
        static_scene_.DeIntegrateLocalMap(currentLocalMap);
        iter = erase(iter);
        n_culled++;
      } else {
        ++iter;
      }
    }
    if (culled) *culled = n_culled;
    return (int)slots.size();
  }

 private:
  int slotFor(double timestamp) {
    Map::iterator it = db_.find(timestamp);
    if (it != db_.end()) return it->second.slot;  // operator[] overwrite of an existing keyframe
    if (free_.empty()) throw std::runtime_error("FusionFrameDataBase: frame store full (raise capacity or enable slide_window)");
    const int slot = free_.back();
    free_.pop_back();
    return slot;
  }
  Map::iterator erase(Map::iterator it) {
    free_.push_back(it->second.slot);
    return db_.erase(it);
  }

  dslam_engine *eng_;
  dslam_frame_store *store_;
  Map db_;
  std::vector<int> free_;
};

}  // namespace SparsetoDense
