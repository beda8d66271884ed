"""MI355X-native TSDF fusion + raycast engine for DenseSLAM-Global-Consistency-h (import name: dslam_amd).

The product is csrc/libdslam_fusion.so: hand-written HIP kernels for gfx950 behind the C ABI of
include/dslam_fusion.h, with ITMLib-compatible C++ classes on top (itmlib/).  This Python package is only a
ctypes view of that ABI for the test-suite, bench.py and harness scripts; it never computes anything itself
and never falls back to a CPU path: opening an engine without the built library or without a GPU raises.

The directory name has hyphens, so load it through ``__graft_entry__.load_package()`` (module ``dslam_amd``).
"""
import os
import subprocess

from ._capi import (BLOCK_SIZE3, HASH_ENTRY_DTYPE, IMAGE_COLOUR_FROM_NORMAL, IMAGE_COLOUR_FROM_VOLUME, IMAGE_DEPTH,
                    IMAGE_SHADED, VOXEL_DTYPE, CApi, DslamError, SceneParams, Stats, TrackerParams, TrackerResult,
                    WeightParams, mat_to_abi)

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(CSRC_DIR, "libdslam_fusion.so")


def build(verbose=False):
    """Compile csrc/*.hip for gfx950 into csrc/libdslam_fusion.so (hipcc cross-compiles without a GPU)."""
    res = subprocess.run(["make", "-C", CSRC_DIR, "-j4"], capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise DslamError("building libdslam_fusion.so failed")
    return LIB_PATH


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7 + libhsa-runtime64; two HSA runtimes in one process
    cannot both own the GPU ("No HIP GPUs are available" from whichever comes second).  Python callers of this
    package use torch for device buffers and RCCL, so bind the engine to the runtime torch will use: load torch's
    bundled HIP runtime first (same SONAME, so libdslam_fusion.so resolves to it) -- without importing torch."""
    import ctypes
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return  # its runtime is already the loaded libamdhip64.so.7
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    bundled = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(bundled):
        ctypes.CDLL(bundled, mode=ctypes.RTLD_GLOBAL)


def open_engine(device=0):
    """Create a HIP engine on `device`.  Raises DslamError when the library or the GPU is missing."""
    _share_torch_hip_runtime()
    return CApi(LIB_PATH, "dslam_", has_engine_device=True, device=device)


def exported_symbols():
    """Names the built library exports (used by the no-GPU ABI test)."""
    out = subprocess.run(["nm", "-D", "--defined-only", LIB_PATH], capture_output=True, text=True, check=True).stdout
    return {line.split()[-1] for line in out.splitlines() if line.strip()}
