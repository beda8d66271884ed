"""ctypes view of the C ABI declared in include/dslam_fusion.h.

`CApi(path, prefix)` binds one shared library that exports the dslam_fusion.h entry points under a symbol
prefix.  The product binds libdslam_fusion.so with prefix ``dslam_`` (see __init__.py); the test suite binds
the CPU oracle with prefix ``oracle_`` through the same class so parity tests drive both identically.
Nothing in this file computes anything: it converts numpy arrays to pointers and back.

Matrix convention: Python callers pass ordinary 4x4 arrays indexed M[row, col]; the ABI wants
ORUtils::Matrix4f storage, column-major float[16] (reference: InfiniTamDriver.cpp:208-226), i.e. M.T.ravel().
"""
import ctypes as C
import os

import numpy as np

HASH_ENTRY_DTYPE = np.dtype(
    {"names": ["pos", "_pad", "offset", "ptr"], "formats": [("<i2", 3), "<i2", "<i4", "<i4"], "offsets": [0, 6, 8, 12],
     "itemsize": 16})
VOXEL_DTYPE = np.dtype(
    {"names": ["sdf", "w_depth", "clr", "w_color", "_pad"], "formats": ["<i2", "u1", ("u1", 3), "u1", "u1"],
     "offsets": [0, 2, 3, 6, 7], "itemsize": 8})

BLOCK_SIZE3 = 512
IMAGE_SHADED, IMAGE_COLOUR_FROM_VOLUME, IMAGE_COLOUR_FROM_NORMAL, IMAGE_DEPTH = 0, 1, 2, 3


class SceneParams(C.Structure):
    """dslam_scene_params (ITMSceneParams + pool sizes)."""
    _fields_ = [("voxel_size", C.c_float), ("mu", C.c_float), ("max_w", C.c_int32), ("frustum_min", C.c_float),
                ("frustum_max", C.c_float), ("stop_integrating_at_max_w", C.c_int32),
                ("num_local_blocks", C.c_int32), ("num_buckets", C.c_int32), ("num_excess", C.c_int32),
                ("use_swapping", C.c_int32), ("history_words", C.c_int32)]

    def __init__(self, voxel_size=0.005, mu=0.02, max_w=100, frustum_min=0.2, frustum_max=3.0,
                 stop_integrating_at_max_w=0, num_local_blocks=0, num_buckets=0, num_excess=0, use_swapping=0,
                 history_words=0):
        super().__init__(voxel_size, mu, max_w, frustum_min, frustum_max, stop_integrating_at_max_w,
                         num_local_blocks, num_buckets, num_excess, use_swapping, history_words)


class TrackerParams(C.Structure):
    """dslam_tracker_params (ITMLibSettings' depth-tracker fields; upstream defaults)."""
    _fields_ = [("no_hierarchy_levels", C.c_int32), ("no_icp_run_till_level", C.c_int32), ("dist_thresh", C.c_float),
                ("termination_threshold", C.c_float), ("regime", C.c_int32 * 8)]

    def __init__(self, levels=5, run_till_level=0, dist_thresh=0.1 * 0.1, termination_threshold=1e-3, regime=None):
        regime = list(regime) if regime is not None else [3, 3, 1, 1, 1]
        regime = (regime + [4] * 8)[:8]
        super().__init__(levels, run_till_level, dist_thresh, termination_threshold, (C.c_int32 * 8)(*regime))


class TrackerResult(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("valid_points_last", C.c_int32), ("f_last", C.c_float), ("pad", C.c_int32)]


class WeightParams(C.Structure):
    _fields_ = [("depth_weighting", C.c_int32), ("max_new_w", C.c_int32), ("max_distance", C.c_float)]


class Stats(C.Structure):
    _fields_ = [("num_allocated_blocks", C.c_int32), ("last_free_block_id", C.c_int32),
                ("last_free_excess_id", C.c_int32), ("no_visible_entries", C.c_int32),
                ("decayed_block_count", C.c_int64), ("slid_block_count", C.c_int64), ("frame_counter", C.c_int32),
                ("fusion_fifo_len", C.c_int32), ("defusion_fifo_len", C.c_int32), ("alloc_failures", C.c_int32),
                ("last_swapped_in", C.c_int32), ("last_swapped_out", C.c_int32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class DslamError(RuntimeError):
    pass


def mat_to_abi(M):
    """4x4 math matrix (M[row, col]) -> column-major float32[16] as the ABI expects."""
    M = np.asarray(M, dtype=np.float32)
    if M.shape != (4, 4):
        raise ValueError("expected a 4x4 matrix")
    return np.ascontiguousarray(M.T).ravel()


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _vptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class _Handle:
    def __init__(self, api, ptr, destroy):
        self.api, self.ptr, self._destroy = api, ptr, destroy

    def close(self):
        if self.ptr:
            self._destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scene(_Handle):
    pass


class RenderState(_Handle):
    width = 0
    height = 0


class View(_Handle):
    pass


class FrameStore(_Handle):
    pass


class CApi:
    """One bound library.  Also plays the role of the engine handle (one engine per CApi instance)."""

    def __init__(self, path, prefix, has_engine_device=True, device=0):
        if not os.path.exists(path):
            raise DslamError(f"shared library not found: {path} (run __graft_entry__.build())")
        self.lib = C.CDLL(path)
        self.prefix = prefix
        self.path = path
        self._engine = C.c_void_p()
        self._pinned = {}
        if has_engine_device:
            self._check(self._fn("engine_create")(C.c_int(device), C.byref(self._engine)), "engine_create")
        else:
            self._check(self._fn("engine_create")(C.byref(self._engine)), "engine_create")

    # -- plumbing --------------------------------------------------------------------------------------
    def _fn(self, name):
        f = getattr(self.lib, self.prefix + name)
        f.restype = C.c_int
        return f

    def has(self, name):
        return hasattr(self.lib, self.prefix + name)

    def _check(self, rc, what):
        if rc < 0:
            msg = ""
            if hasattr(self.lib, self.prefix + "last_error"):
                g = getattr(self.lib, self.prefix + "last_error")
                g.restype = C.c_char_p
                msg = (g() or b"").decode()
            raise DslamError(f"{self.prefix}{what} failed with status {rc} {msg}")
        return rc

    def _call(self, name, *args):
        return self._check(self._fn(name)(*args), name)

    def close(self):
        if self._engine:
            self._fn("engine_destroy")(self._engine)
            self._engine = C.c_void_p()

    # -- engine ----------------------------------------------------------------------------------------
    def set_async(self, flag):
        self._call("engine_set_async", self._engine, C.c_int(int(flag)))

    def device_numa_node(self, device=0):
        """NUMA node of the host the device hangs off (-1: unknown) -- HIP engine"""
        node = C.c_int(-1)
        self._call("device_numa_node", C.c_int(int(device)), C.byref(node))
        return node.value

    def selftest_division(self, samples):
        out = C.c_longlong(-1)
        self._call("selftest_division", self._engine, C.c_longlong(samples), C.byref(out))
        return out.value

    def debug_set_render_tile_budget(self, budget):
        self._call("debug_set_render_tile_budget", self._engine, C.c_int(budget))

    def synchronize(self):
        self._call("engine_synchronize", self._engine)

    # fences: markers in the engine's stream for callers that pipeline frames across PCIe (async mode)
    def fence_create(self):
        h = C.c_void_p()
        self._call("fence_create", self._engine, C.byref(h))
        return _Handle(self, h, self._fn("fence_destroy"))

    def fence_record(self, fence):
        self._call("fence_record", self._engine, fence.ptr)

    def fence_wait(self, fence):
        self._call("fence_wait", fence.ptr)

    def fence_query(self, fence):
        done = C.c_int(0)
        self._call("fence_query", fence.ptr, C.byref(done))
        return bool(done.value)

    def stream(self):
        f = getattr(self.lib, self.prefix + "engine_stream")
        f.restype = C.c_void_p
        return f(self._engine)

    def set_threads(self, n):  # oracle only
        self._call("engine_set_threads", self._engine, C.c_int(n))

    def max_threads(self):  # oracle only
        return self._fn("max_threads")()

    def set_fusion_weight_params(self, depth_weighting=False, max_new_w=1, max_distance=1.0):
        w = WeightParams(int(depth_weighting), int(max_new_w), float(max_distance))
        self._call("set_fusion_weight_params", self._engine, C.byref(w))

    # -- objects ---------------------------------------------------------------------------------------
    def create_scene(self, params, ext_voxel_blocks_dev=None):
        h = C.c_void_p()
        if self.prefix == "dslam_":
            self._call("scene_create", self._engine, C.byref(params), C.c_void_p(ext_voxel_blocks_dev or 0), C.byref(h))
        else:
            self._call("scene_create", self._engine, C.byref(params), C.byref(h))
        s = Scene(self, h, self._fn("scene_destroy"))
        p = SceneParams()
        self._call("scene_get_params", h, C.byref(p))
        s.params = p
        s.n_entries = p.num_buckets + p.num_excess
        return s

    def reset_scene(self, scene):
        self._call("scene_reset", self._engine, scene.ptr)

    def set_shard(self, scene, shard, num_shards, chunk_blocks=256):
        self._call("scene_set_shard", scene.ptr, C.c_int(shard), C.c_int(num_shards), C.c_int(chunk_blocks))

    def set_shard_range(self, scene, first_block, num_blocks):
        self._call("scene_set_shard_range", scene.ptr, C.c_int(first_block), C.c_int(num_blocks))

    def create_render_state(self, scene, width, height):
        h = C.c_void_p()
        self._call("render_state_create", self._engine, scene.ptr, C.c_int(width), C.c_int(height), C.byref(h))
        r = RenderState(self, h, self._fn("render_state_destroy"))
        r.width, r.height, r.n_entries, r.n_local = width, height, scene.n_entries, scene.params.num_local_blocks
        return r

    def create_view(self, width, height, width_d=None, height_d=None):
        h = C.c_void_p()
        width_d = width if width_d is None else width_d
        height_d = height if height_d is None else height_d
        self._call("view_create", self._engine, C.c_int(width), C.c_int(height), C.c_int(width_d), C.c_int(height_d),
                   C.byref(h))
        v = View(self, h, self._fn("view_destroy"))
        v.width, v.height, v.width_d, v.height_d = width, height, width_d, height_d
        return v

    # -- view ------------------------------------------------------------------------------------------
    def view_update(self, view, rgba, depth_mm, affine_a=1.0 / 1000.0, affine_b=0.0, timestamp=0.0, bilateral=False):
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        depth_mm = np.ascontiguousarray(depth_mm, dtype=np.int16)
        assert rgba.size == view.width * view.height * 4 and depth_mm.size == view.width_d * view.height_d
        self._call("view_update", self._engine, view.ptr, _vptr(rgba), _vptr(depth_mm), C.c_float(affine_a),
                   C.c_float(affine_b), C.c_double(timestamp), C.c_int(int(bilateral)))

    def view_update_device(self, view, rgba_dev_ptr, depth_dev_ptr, affine_a=1.0 / 1000.0, affine_b=0.0,
                           timestamp=0.0, bilateral=False):
        self._call("view_update_device", self._engine, view.ptr, C.c_void_p(rgba_dev_ptr), C.c_void_p(depth_dev_ptr),
                   C.c_float(affine_a), C.c_float(affine_b), C.c_double(timestamp), C.c_int(int(bilateral)))

    def view_update_bgr(self, view, bgr, depth_mm, affine_a=1.0 / 1000.0, affine_b=0.0, timestamp=0.0, bilateral=False):
        """CvToItm + UpdateView: packed OpenCV BGR (H, W, 3) in, converted to RGBA on the device."""
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        depth_mm = np.ascontiguousarray(depth_mm, dtype=np.int16)
        assert bgr.size == view.width * view.height * 3 and depth_mm.size == view.width_d * view.height_d
        self._call("view_update_bgr", self._engine, view.ptr, _vptr(bgr), _vptr(depth_mm), C.c_float(affine_a),
                   C.c_float(affine_b), C.c_double(timestamp), C.c_int(int(bilateral)))

    def view_update_bgr_device(self, view, bgr_dev_ptr, depth_dev_ptr, affine_a=1.0 / 1000.0, affine_b=0.0,
                               timestamp=0.0, bilateral=False):
        self._call("view_update_bgr_device", self._engine, view.ptr, C.c_void_p(bgr_dev_ptr), C.c_void_p(depth_dev_ptr),
                   C.c_float(affine_a), C.c_float(affine_b), C.c_double(timestamp), C.c_int(int(bilateral)))

    # -- keyframe store ----------------------------------------------------------------------------------
    def create_frame_store(self, width, height, capacity, width_d=None, height_d=None):
        h = C.c_void_p()
        width_d = width if width_d is None else width_d
        height_d = height if height_d is None else height_d
        self._call("frame_store_create", self._engine, C.c_int(width), C.c_int(height), C.c_int(width_d),
                   C.c_int(height_d), C.c_int(capacity), C.byref(h))
        fs = FrameStore(self, h, self._fn("frame_store_destroy"))
        fs.width, fs.height, fs.width_d, fs.height_d, fs.capacity = width, height, width_d, height_d, capacity
        return fs

    def frame_store_put(self, fs, slot, rgba, depth_mm):
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        depth_mm = np.ascontiguousarray(depth_mm, dtype=np.int16)
        assert rgba.size == fs.width * fs.height * 4 and depth_mm.size == fs.width_d * fs.height_d
        self._call("frame_store_put", self._engine, fs.ptr, C.c_int(slot), _vptr(rgba), _vptr(depth_mm))

    def frame_store_put_bgr(self, fs, slot, bgr, depth_mm):
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        depth_mm = np.ascontiguousarray(depth_mm, dtype=np.int16)
        assert bgr.size == fs.width * fs.height * 3 and depth_mm.size == fs.width_d * fs.height_d
        self._call("frame_store_put_bgr", self._engine, fs.ptr, C.c_int(slot), _vptr(bgr), _vptr(depth_mm))

    def frame_store_put_view(self, fs, slot, view):
        self._call("frame_store_put_view", self._engine, fs.ptr, C.c_int(slot), view.ptr)

    def frame_store_get(self, fs, slot):
        rgba = np.empty((fs.height, fs.width, 4), dtype=np.uint8)
        depth = np.empty((fs.height_d, fs.width_d), dtype=np.int16)
        self._call("frame_store_get", self._engine, fs.ptr, C.c_int(slot), _vptr(rgba), _vptr(depth))
        return rgba, depth

    def frame_store_enable_lists(self, fs, scene):
        self._call("frame_store_enable_lists", self._engine, fs.ptr, scene.ptr)

    def frame_store_put_visible_list(self, fs, slot, scene, rs):
        """Keep the render state's visible list (the blocks the keyframe was just fused into) with the keyframe."""
        self._call("frame_store_put_visible_list", self._engine, fs.ptr, C.c_int(slot), scene.ptr, rs.ptr)

    def deprocess_frame_stored(self, scene, view, fs, slot, M_d, intr, M_rgb=None, intr_rgb=None):
        """DeProcessFrame on exactly the blocks stored with the keyframe (no allocation pass; render state untouched)."""
        m, k = self._mi(M_d, intr)
        mr = mat_to_abi(M_rgb) if M_rgb is not None else None
        kr = np.ascontiguousarray(intr_rgb, dtype=np.float32) if intr_rgb is not None else None
        self._call("deprocess_frame_stored", self._engine, scene.ptr, view.ptr, fs.ptr, C.c_int(slot), _fptr(m), _fptr(k),
                   _fptr(mr), _fptr(kr))

    def reintegrate_batch(self, scene, view, rs, fs, slots, old_Ms, new_Ms, intr, affine_a=1.0 / 1000.0, affine_b=0.0):
        """DenseSlam::OnlineCorrection's loop (DenseSlam.cpp:389-403) over keyframes of a store as ONE call: equal to
        view_update_from_store + deprocess_frame_stored(old pose) + process_frame(new pose, is_defusion) +
        frame_store_put_visible_list per keyframe; the HIP engine runs it block-major (each touched block loaded once)."""
        n = len(slots)
        sl = np.ascontiguousarray(slots, dtype=np.int32)
        om = np.ascontiguousarray(np.stack([mat_to_abi(m) for m in old_Ms]) if n else np.zeros((0, 16)), dtype=np.float32)
        nm = np.ascontiguousarray(np.stack([mat_to_abi(m) for m in new_Ms]) if n else np.zeros((0, 16)), dtype=np.float32)
        k = np.ascontiguousarray(intr, dtype=np.float32)
        self._call("reintegrate_batch", self._engine, scene.ptr, view.ptr, rs.ptr, fs.ptr, C.c_int(n), _vptr(sl), _fptr(om), _fptr(nm),
                   _fptr(k), C.c_float(affine_a), C.c_float(affine_b))

    def debug_inject_device_error(self, scene, bits):
        self._call("debug_inject_device_error", self._engine, scene.ptr, C.c_int(int(bits)))

    def debug_set_push_job_min(self, n):
        self._call("debug_set_push_job_min", self._engine, C.c_int(int(n)))

    def debug_stream_launches(self):
        n = C.c_longlong(0)
        self._call("debug_stream_launches", self._engine, C.byref(n))
        return n.value

    def reintegrate_batch_stats(self, scene):
        """(blocks the last batch loaded, its block-operations: (block, keyframe) pairs de-integrated or re-fused) -- HIP engine"""
        b, o = C.c_int32(0), C.c_int32(0)
        self._call("reintegrate_batch_stats", self._engine, scene.ptr, C.byref(b), C.byref(o))
        return b.value, o.value

    def scene_table_changed(self, scene):
        """after the hash table was written through dslam_scene_hash_table_dev: rebuild the allocation bitmap (HIP engine)"""
        self._call("scene_table_changed", self._engine, scene.ptr)

    def view_update_from_store(self, view, fs, slot, affine_a=1.0 / 1000.0, affine_b=0.0, timestamp=0.0, bilateral=False):
        self._call("view_update_from_store", self._engine, view.ptr, fs.ptr, C.c_int(slot), C.c_float(affine_a),
                   C.c_float(affine_b), C.c_double(timestamp), C.c_int(int(bilateral)))

    def view_update_dataset(self, view, colour, depth_raw, depth_format, max_depth_m, affine_a=1.0 / 1000.0, affine_b=0.0,
                            timestamp=0.0, bilateral=False):
        """Colour image (H, W, 4) RGBA or (H, W, 3) BGR plus the dataset's raw 16-bit depth image; depth_format:
        0 millimetres, 1 KITTI-style depth * 256, 2 TUM / ICL-NUIM (divide by 5)."""
        colour = np.ascontiguousarray(colour, dtype=np.uint8)
        ch = colour.shape[-1]
        raw = np.ascontiguousarray(depth_raw, dtype=np.int16)
        assert colour.size == view.width * view.height * ch and raw.size == view.width_d * view.height_d
        self._call("view_update_dataset", self._engine, view.ptr, _vptr(colour), C.c_int(ch), _vptr(raw),
                   C.c_int(depth_format), C.c_float(max_depth_m), C.c_float(affine_a), C.c_float(affine_b),
                   C.c_double(timestamp), C.c_int(int(bilateral)))

    def download_view_raw_depth(self, view):
        out = np.empty((view.height_d, view.width_d), dtype=np.int16)
        self._call("download_view_raw_depth", self._engine, view.ptr, _vptr(out))
        return out

    def get_depth_image_int16(self, scene, rs, M, intr, scale):
        """GetImage(FREECAMERA_DEPTH) as int16 = (int16)(metres * scale): scale 1000 -> mm, 256 -> the PNG format."""
        m, k = self._mi(M, intr)
        out = np.empty((rs.height, rs.width), dtype=np.int16)
        self._call("get_depth_image_int16", self._engine, scene.ptr, rs.ptr, _fptr(m), _fptr(k), C.c_int(scale), _vptr(out))
        return out

    def download_view_rgba(self, view):
        out = np.empty((view.height, view.width, 4), dtype=np.uint8)
        self._call("download_view_rgba", self._engine, view.ptr, _vptr(out))
        return out

    def depth_post_processing(self, curr_depth_mm, prev_depth_mm, Tpc, intr, filter_threshold, filter_area):
        """DenseSlam::depthPostProcessing's pixel loop on host images; returns (filtered depth, count)."""
        curr = np.array(curr_depth_mm, dtype=np.int16, order="C")
        prev = np.ascontiguousarray(prev_depth_mm, dtype=np.int16)
        assert curr.ndim == 2 and curr.shape == prev.shape
        m, k = self._mi(Tpc, intr)
        count = C.c_int(0)
        self._call("depth_post_processing", self._engine, _vptr(curr), _vptr(prev), C.c_int(curr.shape[1]),
                   C.c_int(curr.shape[0]), _fptr(m), _fptr(k), C.c_float(filter_threshold), C.c_float(filter_area),
                   C.byref(count))
        return curr, count.value

    def depth_post_processing_device(self, curr_dev_ptr, prev_dev_ptr, width, height, Tpc, intr, filter_threshold,
                                     filter_area, want_count=True):
        m, k = self._mi(Tpc, intr)
        count = C.c_int(0)
        self._call("depth_post_processing_device", self._engine, C.c_void_p(curr_dev_ptr), C.c_void_p(prev_dev_ptr),
                   C.c_int(width), C.c_int(height), _fptr(m), _fptr(k), C.c_float(filter_threshold),
                   C.c_float(filter_area), C.byref(count) if want_count else None)
        return count.value

    # -- fusion ----------------------------------------------------------------------------------------
    @staticmethod
    def _mi(M, intr):
        return mat_to_abi(M), np.ascontiguousarray(intr, dtype=np.float32)

    def allocate_scene_from_depth(self, scene, view, rs, M_d, intr, only_update_visible_list=False):
        m, k = self._mi(M_d, intr)
        self._call("allocate_scene_from_depth", self._engine, scene.ptr, view.ptr, rs.ptr, _fptr(m), _fptr(k),
                   C.c_int(int(only_update_visible_list)))

    def integrate_into_scene(self, scene, view, rs, M_d, intr, M_rgb=None, intr_rgb=None):
        m, k = self._mi(M_d, intr)
        mr = mat_to_abi(M_rgb) if M_rgb is not None else None
        kr = np.ascontiguousarray(intr_rgb, dtype=np.float32) if intr_rgb is not None else None
        self._call("integrate_into_scene", self._engine, scene.ptr, view.ptr, rs.ptr, _fptr(m), _fptr(k), _fptr(mr),
                   _fptr(kr))

    def process_frame(self, scene, view, rs, M_d, intr, M_rgb=None, intr_rgb=None, only_update_visible_list=False,
                      is_defusion=False):
        m, k = self._mi(M_d, intr)
        mr = mat_to_abi(M_rgb) if M_rgb is not None else None
        kr = np.ascontiguousarray(intr_rgb, dtype=np.float32) if intr_rgb is not None else None
        self._call("process_frame", self._engine, scene.ptr, view.ptr, rs.ptr, _fptr(m), _fptr(k), _fptr(mr), _fptr(kr),
                   C.c_int(int(only_update_visible_list)), C.c_int(int(is_defusion)))

    def deprocess_frame(self, scene, view, rs, M_d, intr, M_rgb=None, intr_rgb=None):
        m, k = self._mi(M_d, intr)
        mr = mat_to_abi(M_rgb) if M_rgb is not None else None
        kr = np.ascontiguousarray(intr_rgb, dtype=np.float32) if intr_rgb is not None else None
        self._call("deprocess_frame", self._engine, scene.ptr, view.ptr, rs.ptr, _fptr(m), _fptr(k), _fptr(mr),
                   _fptr(kr))

    def decay(self, scene, rs, max_weight, min_age, force_all_voxels, defusion_part=False):
        name = "decay_defusion_part" if defusion_part else "decay"
        self._call(name, self._engine, scene.ptr, rs.ptr if rs is not None else None, C.c_int(max_weight),
                   C.c_int(min_age), C.c_int(int(force_all_voxels)))

    def slide_window(self, scene, rs, max_age):
        self._call("slide_window", self._engine, scene.ptr, rs.ptr if rs is not None else None, C.c_int(max_age))

    def slide_window_defusion_part(self, scene, rs, max_age, max_size):
        self._call("slide_window_defusion_part", self._engine, scene.ptr, rs.ptr if rs is not None else None,
                   C.c_int(max_age), C.c_int(max_size))

    def swap_in(self, scene, rs):
        self._call("swap_in", self._engine, scene.ptr, rs.ptr if rs is not None else None)

    def swap_out(self, scene, rs):
        self._call("swap_out", self._engine, scene.ptr, rs.ptr)

    def save_to_global_memory(self, scene):
        self._call("save_to_global_memory", self._engine, scene.ptr)

    # -- visualisation ---------------------------------------------------------------------------------
    def find_visible_blocks(self, scene, rs, M, intr):
        m, k = self._mi(M, intr)
        self._call("find_visible_blocks", self._engine, scene.ptr, rs.ptr, _fptr(m), _fptr(k))

    def count_visible_blocks(self, scene, rs, min_id, max_id):
        out = C.c_int()
        self._call("count_visible_blocks", self._engine, scene.ptr, rs.ptr, C.c_int(min_id), C.c_int(max_id),
                   C.byref(out))
        return out.value

    def create_expected_depths(self, scene, rs, M, intr):
        m, k = self._mi(M, intr)
        self._call("create_expected_depths", self._engine, scene.ptr, rs.ptr, _fptr(m), _fptr(k))

    def track_camera(self, view, rs, scene_pose_M, pose_M, intr, params=None):
        """ITMDepthTracker::TrackCamera; returns (tracked world->camera pose as a 4x4 row-major array, result)."""
        params = params or TrackerParams()
        sp, k = self._mi(scene_pose_M, intr)
        pose = mat_to_abi(pose_M).copy()
        res = TrackerResult()
        self._call("track_camera", self._engine, view.ptr, rs.ptr, _fptr(sp), _fptr(pose), _fptr(k), C.byref(params),
                   C.byref(res))
        return pose.reshape(4, 4).T.copy(), res

    def _image_call(self, name, scene, rs, M, intr, image_type, download=True, out=None):
        """`out`: an existing (H, W) float32 / (H, W, 4) uint8 array to fill instead of a fresh one (e.g. a page-locked
        array from host_alloc, which the copy back DMAs into directly)."""
        m, k = self._mi(M, intr)
        out_rgba = out_f = None
        if download or out is not None:
            if image_type == IMAGE_DEPTH:
                out_f = out if out is not None else np.empty((rs.height, rs.width), dtype=np.float32)
                assert out_f.dtype == np.float32 and out_f.size == rs.height * rs.width and out_f.flags.c_contiguous
            else:
                out_rgba = out if out is not None else np.empty((rs.height, rs.width, 4), dtype=np.uint8)
                assert out_rgba.dtype == np.uint8 and out_rgba.size == rs.height * rs.width * 4 and out_rgba.flags.c_contiguous
        self._call(name, self._engine, scene.ptr, rs.ptr, _fptr(m), _fptr(k), C.c_int(image_type), _vptr(out_rgba),
                   _fptr(out_f))
        return out_f if image_type == IMAGE_DEPTH else out_rgba

    def render_image(self, scene, rs, M, intr, image_type, download=True, out=None):
        return self._image_call("render_image", scene, rs, M, intr, image_type, download, out)

    def get_image(self, scene, rs, M, intr, image_type, download=True, out=None):
        return self._image_call("get_image", scene, rs, M, intr, image_type, download, out)

    def create_icp_maps(self, scene, rs, M, intr, download=True):
        """trackingController->Prepare.  download=False leaves the maps on the device (all the depth tracker needs)."""
        m, k = self._mi(M, intr)
        if not download:
            self._call("create_icp_maps", self._engine, scene.ptr, rs.ptr, _fptr(m), _fptr(k), None, None)
            return None, None
        pts = np.empty((rs.height, rs.width, 4), dtype=np.float32)
        nrm = np.empty((rs.height, rs.width, 4), dtype=np.float32)
        self._call("create_icp_maps", self._engine, scene.ptr, rs.ptr, _fptr(m), _fptr(k), _fptr(pts), _fptr(nrm))
        return pts, nrm

    def download_icp_maps(self, rs):
        """The points / normals maps the last create_icp_maps left on the device."""
        pts = np.empty((rs.height, rs.width, 4), dtype=np.float32)
        nrm = np.empty((rs.height, rs.width, 4), dtype=np.float32)
        self._call("download_icp_maps", self._engine, rs.ptr, _fptr(pts), _fptr(nrm))
        return pts, nrm

    def download_raycast_image(self, rs):
        """renderState->raycastImage: the grey tracking raycast the last create_icp_maps drew (what
        ITMMainEngine::GetImage(InfiniTAM_IMAGE_SCENERAYCAST) copies out)."""
        out = np.empty((rs.height, rs.width, 4), dtype=np.uint8)
        self._call("download_raycast_image", self._engine, rs.ptr, _vptr(out))
        return out

    # -- meshing export --------------------------------------------------------------------------------
    def mesh_scene(self, scene, max_triangles=0, colour=False):
        """SaveCurrSceneToMesh's MeshScene: returns (positions [n, 3, 3] float32 in metres, colours [n, 3, 3] float32
        in [0, 1] or None), triangles in upstream's CPU-engine order."""
        n = C.c_int(0)
        self._call("mesh_scene", self._engine, scene.ptr, C.c_int(int(max_triangles)), C.c_int(int(colour)), C.byref(n))
        pos = np.empty((max(n.value, 1), 3, 3), dtype=np.float32)  # never a null pointer, even for an empty mesh
        col = np.empty((max(n.value, 1), 3, 3), dtype=np.float32) if colour else None
        self._call("mesh_download", self._engine, _fptr(pos), _fptr(col) if colour else None, C.c_int(n.value))
        return pos[:n.value], (col[:n.value] if colour else None)

    # -- page-locked host images -----------------------------------------------------------------------
    def host_alloc(self, shape, dtype):
        """A zero-filled numpy array over page-locked memory (dslam_host_alloc): view_update* in synchronous mode
        DMA straight out of such arrays.  Release with host_free(array) once nothing refers to it any more."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        p = C.c_void_p()
        self._call("host_alloc", C.c_size_t(n), C.byref(p))
        buf = (C.c_char * max(n, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def host_free(self, arr):
        self._call("host_free", C.c_void_p(self._pinned.pop(arr.ctypes.data)))

    # -- read-back -------------------------------------------------------------------------------------
    def stats(self, scene, rs=None):
        st = Stats()
        self._call("get_stats", self._engine, scene.ptr, rs.ptr if rs is not None else None, C.byref(st))
        return st.as_dict()

    def download_hash_table(self, scene):
        out = np.empty(scene.n_entries, dtype=HASH_ENTRY_DTYPE)
        self._call("download_hash_table", self._engine, scene.ptr, _vptr(out))
        return out

    def download_voxel_blocks(self, scene, first=0, count=None):
        count = scene.params.num_local_blocks - first if count is None else count
        out = np.empty((count, BLOCK_SIZE3), dtype=VOXEL_DTYPE)
        self._call("download_voxel_blocks", self._engine, scene.ptr, C.c_int(first), C.c_int(count), _vptr(out))
        return out

    def download_allocation_list(self, scene):
        out = np.empty(scene.params.num_local_blocks, dtype=np.int32)
        self._call("download_allocation_list", self._engine, scene.ptr, _vptr(out))
        return out

    def download_excess_list(self, scene):
        out = np.empty(scene.params.num_excess, dtype=np.int32)
        self._call("download_excess_list", self._engine, scene.ptr, _vptr(out))
        return out

    def download_visible_ids(self, rs):
        out = np.empty(rs.n_local, dtype=np.int32)
        n = C.c_int()
        self._call("download_visible_ids", self._engine, rs.ptr, _vptr(out), C.c_int(rs.n_local), C.byref(n))
        return out[:min(n.value, rs.n_local)].copy()

    def download_visible_types(self, rs):
        out = np.empty(rs.n_entries, dtype=np.uint8)
        self._call("download_visible_types", self._engine, rs.ptr, _vptr(out))
        return out

    def download_range_image(self, rs):
        out = np.empty((rs.height, rs.width, 2), dtype=np.float32)
        self._call("download_range_image", self._engine, rs.ptr, _fptr(out))
        return out

    def download_raycast_result(self, rs):
        out = np.empty((rs.height, rs.width, 4), dtype=np.float32)
        self._call("download_raycast_result", self._engine, rs.ptr, _fptr(out))
        return out

    def download_view_depth(self, view):
        out = np.empty((view.height_d, view.width_d), dtype=np.float32)
        self._call("download_view_depth", self._engine, view.ptr, _fptr(out))
        return out

    def download_swap_states(self, scene):
        out = np.empty(scene.n_entries, dtype=np.uint8)
        self._call("download_swap_states", self._engine, scene.ptr, _vptr(out))
        return out

    def download_alloc_scratch(self, scene):
        types = np.empty(scene.n_entries, dtype=np.uint8)
        coords = np.empty((scene.n_entries, 4), dtype=np.int16)
        self._call("download_alloc_scratch", self._engine, scene.ptr, _vptr(types), _vptr(coords))
        return types, coords

    def download_last_seen(self, scene):
        out = np.empty(scene.params.num_local_blocks, dtype=np.int32)
        self._call("download_last_seen", self._engine, scene.ptr, _vptr(out))
        return out

    def download_stored_block(self, scene, entry):
        out = np.empty(BLOCK_SIZE3, dtype=VOXEL_DTYPE)
        rc = self._call("download_stored_block", self._engine, scene.ptr, C.c_int(entry), _vptr(out))
        return bool(rc), out

    def upload_scene_state(self, scene, hash_table=None, allocation_list=None, last_free_block_id=0, excess_list=None,
                           last_free_excess_id=0):
        h = np.ascontiguousarray(hash_table, dtype=HASH_ENTRY_DTYPE) if hash_table is not None else None
        a = np.ascontiguousarray(allocation_list, dtype=np.int32) if allocation_list is not None else None
        x = np.ascontiguousarray(excess_list, dtype=np.int32) if excess_list is not None else None
        self._call("upload_scene_state", self._engine, scene.ptr, _vptr(h), _vptr(a), C.c_int(last_free_block_id),
                   _vptr(x), C.c_int(last_free_excess_id))

    def upload_voxel_blocks(self, scene, first, blocks):
        b = np.ascontiguousarray(blocks, dtype=VOXEL_DTYPE)
        self._call("upload_voxel_blocks", self._engine, scene.ptr, C.c_int(first), C.c_int(b.size // BLOCK_SIZE3),
                   _vptr(b))

    def track_dirty(self, scene, enable):
        self._call("scene_track_dirty", self._engine, scene.ptr, C.c_int(int(enable)))

    def shard_dirty_plan(self, scene, num_shards, chunk_blocks):
        counts = (C.c_int32 * num_shards)()
        self._call("shard_dirty_plan", self._engine, scene.ptr, C.c_int(num_shards), C.c_int(chunk_blocks), counts)
        return list(counts)

    def shard_dirty_pack(self, scene, shard, send_ptr, capacity_blocks):
        self._call("shard_dirty_pack", self._engine, scene.ptr, C.c_int(shard), C.c_void_p(send_ptr), C.c_int(capacity_blocks))

    def shard_dirty_unpack(self, scene, skip_shard, recv_ptr, stride_blocks):
        self._call("shard_dirty_unpack", self._engine, scene.ptr, C.c_int(skip_shard), C.c_void_p(recv_ptr), C.c_int(stride_blocks))

    def upload_visible_ids(self, rs, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        self._call("upload_visible_ids", self._engine, rs.ptr, _vptr(ids), C.c_int(ids.size))

    # -- instrumentation (product only) ----------------------------------------------------------------
    def time_integrate(self, scene, view, rs, M_d, intr, iterations):
        m, k = self._mi(M_d, intr)
        ms = C.c_float()
        nb = C.c_int()
        self._call("time_integrate", self._engine, scene.ptr, view.ptr, rs.ptr, _fptr(m), _fptr(k), C.c_int(iterations),
                   C.byref(ms), C.byref(nb))
        return ms.value, nb.value

    def kernel_timer_enable(self, flag):
        self._call("kernel_timer_enable", self._engine, C.c_int(int(flag)))

    def kernel_timer_read(self):
        ms = C.c_double()
        n = C.c_int64()
        nb = C.c_int64()
        self._call("kernel_timer_read", self._engine, C.byref(ms), C.byref(n), C.byref(nb))
        return ms.value, n.value, nb.value

    def scene_voxel_blocks_dev(self, scene):
        f = getattr(self.lib, self.prefix + "scene_voxel_blocks_dev")
        f.restype = C.c_void_p
        return f(scene.ptr)

    def render_state_image_dev(self, rs, want_float):
        f = getattr(self.lib, self.prefix + "render_state_image_dev")
        f.restype = C.c_void_p
        return f(rs.ptr, C.c_int(int(want_float)))
