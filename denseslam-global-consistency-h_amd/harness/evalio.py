"""Wire formats and metrics on either side of the hot path, so results of this engine
can be compared with the reference's own artefacts:

* raycast-depth dumps: DenseSlam::SaveRaycastDepth writes uint16 = depth_m * 256 (reference
  src/DenseSLAM/InfiniTamDriver.cpp:187-199, DenseSlam.cpp:573-603);
* accuracy metrics of scripts/eval_raycast_depth.py:90-136 (crop bottom-centre 912x228, mask 0.01..50 m, errors in mm);
* the per-frame memory log ../memory.txt (DenseSLAMGUI.cpp:576-595; value = used GiB * 10.24, i.e. units of 100 MiB),
  used bytes = sizeof(ITMVoxel) * 512 * (numAllocatedVoxelBlocks - lastFreeBlockId) (InfiniTamDriver.h:344-347).
"""
import math

import numpy as np


def depth_to_png16(depth_m):
    """FloatDepthmapToInt16: int16(depth * 256), stored in a 16-bit PNG."""
    return (np.asarray(depth_m, np.float32) * np.float32(256)).astype(np.int16)


def png16_to_depth(png):
    """depth_read of the evaluation script: value / 256."""
    return np.asarray(png).astype(np.float64) / 256.0


def crop_bottom_centre(img, width=912, height=228):
    h, w = img.shape[:2]
    i = h - height
    j = int(round((w - width) / 2.0))
    return img[i:i + height, j:j + width]


def depth_metrics(pred_m, gt_m, lo=0.01, hi=50.0):
    """The 12 numbers eval_raycast_depth.py prints for one image pair (errors in millimetres)."""
    pred_m, gt_m = np.asarray(pred_m, np.float64), np.asarray(gt_m, np.float64)
    mask = (pred_m > lo) & (gt_m > lo) & (pred_m < hi) & (gt_m < hi)
    out_mm, tgt_mm = 1e3 * pred_m[mask], 1e3 * gt_m[mask]
    n = int(mask.sum())
    if n == 0:
        return {"mask_number": 0}
    ad = np.abs(out_mm - tgt_mm)
    ratio = np.maximum(out_mm / tgt_mm, tgt_mm / out_mm)
    res = {"mask_number": n, "mae": float(ad.mean()), "rmse": math.sqrt(float((ad * ad).mean())),
           "lg10": float(np.abs(np.log10(out_mm) - np.log10(tgt_mm)).mean()), "absrel": float((ad / tgt_mm).mean()),
           "squared_rel": float(((ad / tgt_mm) ** 2).mean())}
    for k in (1, 2, 3):
        res[f"delta{k}_125"] = float((ratio < 1.25 ** k).mean())
        res[f"delta{k}_101"] = float((ratio < 1.01 ** k).mean())
    return res


def used_memory_bytes(stats, voxel_bytes=8):
    """InfiniTamDriver::GetLocalMapUsedMemoryBytes (InfiniTamDriver.h:344-347) from dslam_stats."""
    return voxel_bytes * 512 * (stats["num_allocated_blocks"] - stats["last_free_block_id"])


def memory_log_line(frame_no, used_bytes):
    """One line of ../memory.txt: '<frame> <used GiB * 10.24>' with the C++ stream's 6 significant digits."""
    value = np.float32(np.float32(used_bytes) * np.float32(1.0 / 1024.0 / 1024.0 / 1024.0)) * np.float32(10.24)
    return "%d %s" % (frame_no, ("%.6g" % float(value)))
