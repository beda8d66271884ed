"""The reference's only recorded behaviour of its memory path: ../memory*.txt, one "<frame> <used GiB * 10.24>" line per GUI
tick (DenseSLAMGUI.cpp:576-595), for the four settings scripts/memoryDraw.py:12-13 plots --

    memory.txt                    "Origin"                              voxel_decay 0, slide_window 0
    memory_decay.txt              "Map regularization"                  voxel_decay 1, slide_window 0
    memory_slide_window.txt       "Slidewindow"                         voxel_decay 0, slide_window 1
    memory_decay_slide_window.txt "Slidewindow + Map regularization"    voxel_decay 1, slide_window 1

(param.yaml keys voxel_decay / min_decay_age / max_decay_weight / slide_window / max_age, SystemEntry.cpp:138-149).  This
module replays DenseSlam::ProcessFrame's per-keyframe calls (DenseSlam.cpp:210-232: IntegrateLocalMap, SlideWindow once the
fusion database holds more than max_age keyframes, Decay) on any engine, writes logs in that format, and reduces a set of
four curves to the SHAPE numbers that can be compared across datasets (tests/golden/reference_memory_shape.json holds the
same numbers extracted from the reference's own four files).

    python denseslam-global-consistency-h_amd/harness/memory_curves.py [keyframes] [out_dir]
"""
import json
import os
import sys

import numpy as np

MODES = {"memory": (0, 0), "memory_decay": (1, 0), "memory_slide_window": (0, 1), "memory_decay_slide_window": (1, 1)}
POOL_UNITS = 10.24  # a 0x40000-block pool in the log's unit (GiB * 10.24)


def run_mode(api, pkg, wl, params, frames, voxel_decay, min_decay_age, max_decay_weight, slide_window, max_age, force_all=True):
    """Used bytes after every keyframe (InfiniTamDriver::GetLocalMapUsedMemoryBytes, InfiniTamDriver.h:344-347) until the
    pool cannot serve an allocation any more (the reference's "Origin" and "Map regularization" logs end there)."""
    from dslam_amd.harness import evalio
    scene = api.create_scene(params)
    rs = api.create_render_state(scene, wl.W, wl.H)
    view = api.create_view(wl.W, wl.H)
    used, exhausted_at = [], None
    for i, (rgba, mm, M) in enumerate(frames):
        api.view_update(view, rgba, mm, timestamp=float(i))
        api.process_frame(scene, view, rs, M, wl.intr)                      # DenseSlam.cpp:213
        st = api.stats(scene, rs)
        if slide_window and st["fusion_fifo_len"] > max_age:                # :215-225 (database size > max_age)
            api.slide_window(scene, rs, max_age)
        if voxel_decay:                                                     # :227-232 (forceAllVoxels = true, InfiniTamDriver.h:280)
            api.decay(scene, rs, max_decay_weight, min_decay_age, force_all)
        st2 = api.stats(scene, rs)
        st2["num_allocated_blocks"] = scene.params.num_local_blocks
        used.append(evalio.used_memory_bytes(st2))
        if st["alloc_failures"] > 0:
            exhausted_at = i + 1
            break
    for o in (view, rs, scene):
        o.close()
    return used, exhausted_at


def to_units(used_bytes):
    return np.asarray(used_bytes, np.float64) / 2.0 ** 30 * 10.24


def shape_metrics(curves, exhausted, pool_units=POOL_UNITS):
    """curves: name -> per-keyframe values in the log's unit; exhausted: name -> keyframe the pool ran out at (or None).
    Everything is a ratio, so that a dataset with another growth rate per keyframe can be compared."""
    c = {k: np.asarray(v, np.float64) for k, v in curves.items()}
    sw, both = c["memory_slide_window"], c["memory_decay_slide_window"]
    n = min(len(sw), len(both))

    def when_reaches(v, level):
        idx = np.nonzero(v >= level)[0]
        return int(idx[0]) + 1 if len(idx) else None

    full = 0.99 * pool_units
    t_origin = exhausted.get("memory") or when_reaches(c["memory"], full)
    t_decay = exhausted.get("memory_decay") or when_reaches(c["memory_decay"], full)
    # the window is full where the windowed curve leaves the un-windowed one: first keyframe at which it is 3 % below
    m = min(len(sw), len(c["memory"]))
    leave = np.nonzero(sw[:m] < 0.97 * c["memory"][:m])[0]
    t_window = int(leave[0]) + 1 if len(leave) else None
    out = {
        "origin_fills_pool_at": t_origin, "decay_fills_pool_at": t_decay,
        "decay_over_origin_fill_time": (t_decay / t_origin) if (t_origin and t_decay) else None,
        # growth per keyframe over the common early stretch (before anything saturates): decay's share of the origin's
        "decay_over_origin_growth": None,
        "window_full_at": t_window,
        "window_full_over_origin_fill_time": (t_window / t_origin) if (t_window and t_origin) else None,
        "windowed_modes_complete": bool(exhausted.get("memory_slide_window") is None and exhausted.get("memory_decay_slide_window") is None),
        "window_peak_over_pool": float(sw.max() / pool_units), "window_decay_peak_over_pool": float(both.max() / pool_units),
    }
    if t_origin:
        a, b = int(0.15 * t_origin), int(0.6 * t_origin)
        go = (c["memory"][b] - c["memory"][a]) / (b - a)
        gd = (c["memory_decay"][b] - c["memory_decay"][a]) / (b - a)
        out["decay_over_origin_growth"] = float(gd / go)
    if t_window:
        lo = min(n - 1, int(1.05 * t_window))
        ratio = both[lo:n] / sw[lo:n]
        out["window_decay_over_window_after_window_full"] = {"median": float(np.median(ratio)), "min": float(ratio.min()), "max": float(ratio.max())}
        out["window_decay_over_window_at_end"] = float(both[n - 1] / sw[n - 1])
    return out


def write_logs(out_dir, curves_bytes):
    from dslam_amd.harness import evalio
    os.makedirs(out_dir, exist_ok=True)
    for name, used in curves_bytes.items():
        with open(os.path.join(out_dir, name + ".txt"), "w") as f:
            for i, b in enumerate(used):
                f.write(evalio.memory_log_line(i + 1, b) + "\n")


def run_all(api, pkg, wl, params, frames, min_decay_age, max_decay_weight, max_age):
    curves, exhausted = {}, {}
    for name, (vd, sw) in MODES.items():
        curves[name], exhausted[name] = run_mode(api, pkg, wl, params, frames, vd, min_decay_age, max_decay_weight, sw, max_age)
    return curves, exhausted


_WL = None  # the workload, inherited by the forked workers (its trajectory is a closure: not picklable)


def _frame(i):
    return _WL.frame(i)


def generate(wl, n, workers):
    """wl.frame(0..n) on a worker pool (call before this process touches the GPU: the pool forks)."""
    global _WL
    import multiprocessing as mp
    if workers <= 1:
        return [wl.frame(i) for i in range(n)]
    _WL = wl
    with mp.get_context("fork").Pool(workers) as pool:
        return pool.map(_frame, range(n), chunksize=max(1, n // (workers * 4)))


def main():
    sys.path.insert(0, ".")
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from dslam_amd.harness import synth
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    out_dir = sys.argv[2] if len(sys.argv) > 2 else None
    a = dict(min_decay_age=int(os.environ.get("MIN_DECAY_AGE", "30")), max_decay_weight=int(os.environ.get("MAX_DECAY_WEIGHT", "3")),
             max_age=int(os.environ.get("MAX_AGE", "300")))
    noise, outl = float(os.environ.get("STEREO_NOISE_PX", "0")), float(os.environ.get("OUTLIER_FRAC", "0"))
    wl = synth.s_street(640, 480, stereo_noise_px=noise, outlier_frac=outl)
    frames = generate(wl, n, 16)
    eng = pkg.open_engine(0)
    params = pkg.SceneParams(history_words=(a["max_age"] + 64) // 64 + 1, **wl.scene_kwargs)  # default 0x40000-block pool
    curves, exhausted = run_all(eng, pkg, wl, params, frames, **a)
    units = {k: to_units(v) for k, v in curves.items()}
    if out_dir:
        write_logs(out_dir, curves)
    print(json.dumps({"workload": f"{wl.name} 640x480, {n} keyframes, default pool, disparity noise {noise} px, outliers {outl}", "params": a,
                      "keyframes_logged": {k: len(v) for k, v in curves.items()}, "pool_exhausted_at": exhausted,
                      "shape": shape_metrics(units, exhausted)}))


if __name__ == "__main__":
    main()
