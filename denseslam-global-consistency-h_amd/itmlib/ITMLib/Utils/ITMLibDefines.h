// ITMLibDefines.h -- compile-time names the reference uses (InfiniTamDriver.h:333-351: sizeof(ITMVoxel),
// SDF_BLOCK_SIZE3; typedefs ITMVoxel / ITMVoxelIndex at InfiniTamDriver.h:240,362).
#pragma once
#include <string>
#include "../../ORUtils/ORUtils.h"
extern "C" {
#include "dslam_fusion.h"
}

#define SDF_BLOCK_SIZE DSLAM_BLOCK_SIZE
#define SDF_BLOCK_SIZE3 DSLAM_BLOCK_SIZE3
#define SDF_LOCAL_BLOCK_NUM DSLAM_DEFAULT_LOCAL_BLOCK_NUM
#define SDF_BUCKET_NUM DSLAM_DEFAULT_BUCKET_NUM
#define SDF_EXCESS_LIST_SIZE DSLAM_DEFAULT_EXCESS_LIST_SIZE
#define SDF_TRANSFER_BLOCK_NUM DSLAM_TRANSFER_BLOCK_NUM

namespace ITMLib {
namespace Objects {
/// ITMVoxel_s_rgb: 8 bytes, identical to dslam_voxel
struct ITMVoxel_s_rgb {
  short sdf;
  uchar w_depth;
  Vector3u clr;
  uchar w_color;
  static const bool hasColorInformation = true;
  ITMVoxel_s_rgb() : sdf(32767), w_depth(0), clr((uchar)0), w_color(0) {}
};
class ITMVoxelBlockHash;
}  // namespace Objects
}  // namespace ITMLib
typedef ITMLib::Objects::ITMVoxel_s_rgb ITMVoxel;
typedef ITMLib::Objects::ITMVoxelBlockHash ITMVoxelIndex;
static_assert(sizeof(ITMVoxel) == sizeof(dslam_voxel), "ITMVoxel must stay 8 bytes");

namespace ITMLib {
inline void dslam_check(int status, const char *what) {
  if (status < 0) throw std::runtime_error(std::string(what) + ": " + dslam_last_error());
}
}  // namespace ITMLib
