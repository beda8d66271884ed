"""The decay / sliding-window semantics of this build are design decisions (the fork's bodies are not
in the reference tree).  The only thing the reference records about them is four memory logs (memory*.txt, labelled by
scripts/memoryDraw.py:12-13).  These tests replay DenseSlam's per-keyframe calls (DenseSlam.cpp:210-232) in the same four
settings and compare the SHAPE of the resulting curves with the numbers extracted from the reference's files
(tests/golden/reference_memory_shape.json, made by tests/golden/make_reference_memory_shape.py).

What can and cannot be compared.  The logs come from KITTI 2011_09_30_drive_0033 with the fork's own (unknown) yaml; the
dataset is not available, so the input is the synthetic S-street drive with KITTI-like stereo noise (0.5 px disparity
noise, 0.3 % mismatches: without noise nearly every block is observed often and decay has next to nothing to remove --
growth ratio 0.92 instead of the reference's 0.76).  With it, the un-windowed map grows by ~1,560 blocks per keyframe
(reference: 1,477) and the default 0x40000-block pool fills after 119 keyframes (reference: 177 keyframes = 320 frames).
`max_age` is a free yaml key: it is set so that the window fills at the same fraction of the origin's fill time as in the
reference (0.81).  Compared, with tolerances that reflect how much the reference's own curves wander with scene content
(its window+decay / window ratio moves between 0.66 and 0.76 along one run):

  * decay fills the pool later than the origin:      time ratio      reference 1.38   accepted 1.20 .. 1.60
  * window + decay plateau over window-only plateau:  median ratio    reference 0.716  accepted 0.62 .. 0.85
  * both un-windowed settings exhaust the pool, both windowed settings run the whole sequence
  * origin >= decay, window >= window + decay at every keyframe

NOT asserted (waived): the peak of the window-only run over the pool -- 0.960 in the reference's log,
0.84 .. 0.88 here.  It is set by how much of the pool one window's worth of keyframes covers (camera speed, scene geometry of
the drive), and no setting of the two free knobs (noise level, max_age) that keeps the figures above inside their tolerances
reaches it (profiles/r03_memory_shape_sensitivity.json).  The same sweep shows what this pin can tell: it excludes the
degenerate readings of Decay (no age gate: the map never accumulates; nothing decays: ratio 0.93) but NOT the choice between
the gated full sweep and the aged-list mode, which land closer to each other than either knob moves them.
"""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF = json.load(open(os.path.join(HERE, "golden", "reference_memory_shape.json")))["shape"]
NOISE = dict(stereo_noise_px=0.5, outlier_frac=0.003)


def check_shape(shape, curves, fill_tol, ratio_tol):
    assert shape["origin_fills_pool_at"] and shape["decay_fills_pool_at"], "an un-windowed setting did not fill the pool"
    assert shape["windowed_modes_complete"], "a windowed setting ran out of blocks"
    r = shape["decay_over_origin_fill_time"]
    assert fill_tol[0] <= r <= fill_tol[1], f"decay / origin fill time {r:.3f} (reference {REF['decay_over_origin_fill_time']:.3f})"
    m = shape["window_decay_over_window_after_window_full"]["median"]
    assert ratio_tol[0] <= m <= ratio_tol[1], f"window+decay / window {m:.3f} (reference {REF['window_decay_over_window_after_window_full']['median']:.3f})"
    n = min(len(curves["memory"]), len(curves["memory_decay"]))
    assert (np.asarray(curves["memory"][:n]) >= np.asarray(curves["memory_decay"][:n])).all()
    n = min(len(curves["memory_slide_window"]), len(curves["memory_decay_slide_window"]))
    assert (np.asarray(curves["memory_slide_window"][:n]) >= np.asarray(curves["memory_decay_slide_window"][:n])).all()


def test_reference_shape_numbers_are_what_the_logs_say():
    """the golden numbers themselves (extracted from /root/reference/memory*.txt; cross-checked by hand: memory.txt ends
    at line 320 with 10.2151, memory_decay.txt at 442 with 10.2001, the windowed logs run to 1505)"""
    assert REF["origin_fills_pool_at"] == 320 and REF["decay_fills_pool_at"] == 442
    assert abs(REF["decay_over_origin_fill_time"] - 442 / 320) < 1e-9
    assert 0.70 < REF["window_decay_over_window_after_window_full"]["median"] < 0.73
    assert REF["windowed_modes_complete"] and 0.75 < REF["window_full_over_origin_fill_time"] < 0.85


def test_memory_curve_shape_scaled_down_on_the_oracle(pkg, synth, oracle):
    """CPU-sized version: the 640x480 camera at quarter resolution (same field of view, same angular noise), half the
    default pool, decay age 20 -- the same four settings and the same reduction, with wider tolerances.  It pins the
    oracle's semantics to the reference's curve shapes; the full-size run is the GPU test below."""
    from dslam_amd.harness import memory_curves as mc
    wl = synth.s_street(640, 480, stereo_noise_px=NOISE["stereo_noise_px"] / 4, outlier_frac=NOISE["outlier_frac"])
    fx, fy, cx, cy = [float(x) for x in wl.intr]
    wl.W, wl.H, wl.intr = 160, 120, np.array([fx / 4, fy / 4, (cx + 0.5) / 4 - 0.5, (cy + 0.5) / 4 - 0.5], np.float32)
    n = 230
    frames = mc.generate(wl, n, 8)
    params = pkg.SceneParams(num_local_blocks=0x20000, num_buckets=0x40000, num_excess=0x8000, history_words=2, **wl.scene_kwargs)
    oracle.set_threads(min(8, oracle.max_threads()))
    try:
        c0, e0 = mc.run_mode(oracle, pkg, wl, params, frames, 0, 0, 0, 0, 0)
        assert e0 is not None and 60 <= e0 <= 160, f"origin filled the scaled pool at {e0}"
        a = dict(min_decay_age=20, max_decay_weight=3, max_age=int(round(REF["window_full_over_origin_fill_time"] * e0)))
        curves, exhausted = {"memory": c0}, {"memory": e0}
        for name, (vd, sw) in mc.MODES.items():
            if name != "memory":
                curves[name], exhausted[name] = mc.run_mode(oracle, pkg, wl, params, frames, vd, a["min_decay_age"], a["max_decay_weight"], sw, a["max_age"])
    finally:
        oracle.set_threads(1)
    pool_units = params.num_local_blocks * 4096 / 2.0 ** 30 * 10.24
    units = {k: mc.to_units(v) for k, v in curves.items()}
    shape = mc.shape_metrics(units, exhausted, pool_units=pool_units)
    check_shape(shape, units, fill_tol=(1.10, 1.90), ratio_tol=(0.55, 0.92))
    assert len(curves["memory_slide_window"]) == n


@pytest.mark.gpu
def test_memory_curve_shape_full_size(pkg, synth, gpu, tmp_path):
    """1500 keyframes, 640x480, the default 0x40000-block pool, min_decay_age 30 / max_decay_weight 3, in the
    four settings of the reference's logs; logs are written in the reference's format and reduced like the reference's."""
    from dslam_amd.harness import memory_curves as mc
    import subprocess
    import sys
    n = 1500
    wl = synth.s_street(640, 480, **NOISE)
    # (the GPU is initialised in this process, so the frames are rendered by a pool in a process of its own)
    prefix = str(tmp_path / "frames")
    subprocess.run([sys.executable, os.path.join(HERE, "_gen_frames.py"), prefix, str(n), str(min(16, os.cpu_count() or 2)),
                    str(NOISE["stereo_noise_px"]), str(NOISE["outlier_frac"])], check=True, timeout=600)
    rgba, depth, poses = (np.load(prefix + s, mmap_mode="r") for s in ("_rgba.npy", "_depth.npy", "_poses.npy"))
    frames = [(rgba[i], depth[i], poses[i]) for i in range(n)]
    params = pkg.SceneParams(history_words=4, **wl.scene_kwargs)
    c0, e0 = mc.run_mode(gpu, pkg, wl, params, frames[:400], 0, 0, 0, 0, 0)
    assert e0 is not None
    a = dict(min_decay_age=30, max_decay_weight=3, max_age=int(round(REF["window_full_over_origin_fill_time"] * e0)))
    curves, exhausted = {"memory": c0}, {"memory": e0}
    for name, (vd, sw) in mc.MODES.items():
        if name != "memory":
            curves[name], exhausted[name] = mc.run_mode(gpu, pkg, wl, params, frames, vd, a["min_decay_age"], a["max_decay_weight"], sw, a["max_age"])
    mc.write_logs(str(tmp_path), curves)
    assert open(tmp_path / "memory.txt").readline().split()[0] == "1"
    units = {k: mc.to_units(v) for k, v in curves.items()}
    shape = mc.shape_metrics(units, exhausted)
    check_shape(shape, units, fill_tol=(1.20, 1.60), ratio_tol=(0.62, 0.85))
    assert len(curves["memory_slide_window"]) == n and len(curves["memory_decay_slide_window"]) == n
    # growth per keyframe of the un-windowed map: the reference's is 1,477 blocks (0.0577 log units)
    growth_blocks = (units["memory"][e0 - 2] - units["memory"][10]) / (e0 - 12) / 10.24 * 2 ** 30 / 4096
    assert 1000 < growth_blocks < 2200
