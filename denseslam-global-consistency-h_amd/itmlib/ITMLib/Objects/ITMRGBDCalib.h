// ITMRGBDCalib.h -- calibration objects filled by CreateItmCalib (InfiniTamDriver.cpp:55-81) and included directly
// by Input.h:13.
#pragma once
#include "../Utils/ITMLibDefines.h"

namespace ITMLib {
namespace Objects {

class ITMIntrinsics {
 public:
  struct ProjectionParamsSimple {
    Vector4f all;
    float fx, fy, px, py;
  } projectionParamsSimple;
  Vector2i sizeXY;
  void SetFrom(float fx, float fy, float cx, float cy, float sizeX, float sizeY) {
    projectionParamsSimple.fx = fx; projectionParamsSimple.fy = fy;
    projectionParamsSimple.px = cx; projectionParamsSimple.py = cy;
    projectionParamsSimple.all = Vector4f(fx, fy, cx, cy);
    sizeXY = Vector2i((int)sizeX, (int)sizeY);
  }
  ITMIntrinsics() { SetFrom(580, 580, 320, 240, 640, 480); }
};

class ITMExtrinsics {
 public:
  Matrix4f calib, calib_inv;
  void SetFrom(const Matrix4f &src) { calib = src; calib.inv(calib_inv); }
  ITMExtrinsics() { Matrix4f m; m.setIdentity(); SetFrom(m); }
};

class ITMDisparityCalib {
 public:
  typedef enum { TRAFO_KINECT, TRAFO_AFFINE } TrafoType;
  TrafoType type;
  Vector2f params;
  void SetFrom(float a, float b, TrafoType t) { params = Vector2f(a, b); type = t; }
  ITMDisparityCalib() { SetFrom(1.0f / 1000.0f, 0.0f, TRAFO_AFFINE); }
};

class ITMRGBDCalib {
 public:
  ITMIntrinsics intrinsics_rgb, intrinsics_d;
  ITMExtrinsics trafo_rgb_to_depth;
  ITMDisparityCalib disparityCalib;
};

}  // namespace Objects
}  // namespace ITMLib
