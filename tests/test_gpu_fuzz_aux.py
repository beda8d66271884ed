"""Seeded random inputs for the components next to the hot path, HIP engine vs CPU oracle: view pre-processing
(bilateral filter, BGR, dataset depth formats, depthPostProcessing) bit for bit, the depth tracker to 1e-5."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


# DSLAM_FUZZ_AUX_OFFSET=<n> shifts every seed range (one-off wider hunts)
_OFF = int(os.environ.get("DSLAM_FUZZ_AUX_OFFSET", "0"))


@pytest.mark.parametrize("seed", range(_OFF, _OFF + 40))
def test_random_view_preprocessing(pkg, synth, gpu, oracle, seed):
    rng = np.random.default_rng(3000 + seed)
    W, H = int(rng.integers(5, 90)), int(rng.integers(5, 70))
    kind = rng.choice(["smooth", "noise", "holes", "extreme"])
    base = 500 + 3000 * rng.random()
    yy, xx = np.mgrid[0:H, 0:W]
    mm = base + 200 * np.sin(xx / 7.0) + 150 * np.cos(yy / 5.0)
    if kind != "smooth":
        mm = mm + rng.normal(0, 30 if kind == "noise" else 5, (H, W))
    mm = mm.astype(np.int32)
    if kind in ("holes", "extreme"):
        mm[rng.random((H, W)) < 0.2] = 0
    if kind == "extreme":
        mm[rng.random((H, W)) < 0.1] = 32767
        mm[rng.random((H, W)) < 0.05] = -5
        mm[rng.random((H, W)) < 0.05] = 1
    mm = np.clip(mm, -32768, 32767).astype(np.int16)
    rgba = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
    bgr = np.ascontiguousarray(rgba[..., 2::-1])
    fmt = int(rng.integers(0, 3))
    max_m = float(rng.choice([4.0, 10.0, 40.0, 120.0]))
    a, b = (1e-3, 0.0) if rng.random() < 0.7 else (float(rng.uniform(5e-4, 2e-3)), float(rng.uniform(-0.1, 0.1)))
    out = {}
    for name, api in (("gpu", gpu), ("oracle", oracle)):
        v = api.create_view(W, H)
        res = []
        api.view_update(v, rgba, mm, affine_a=a, affine_b=b, bilateral=True)
        res.append(api.download_view_depth(v).view(np.uint32))
        api.view_update_bgr(v, bgr, mm, affine_a=a, affine_b=b)
        res += [api.download_view_depth(v).view(np.uint32), api.download_view_rgba(v)]
        api.view_update_dataset(v, rgba, mm, fmt, max_m, affine_a=a, affine_b=b, bilateral=bool(seed % 2))
        res += [api.download_view_raw_depth(v), api.download_view_depth(v).view(np.uint32)]
        out[name] = res
    for k, (g, o) in enumerate(zip(out["gpu"], out["oracle"])):
        assert np.array_equal(g, o), f"seed {seed}: view output {k} ({kind}, {W}x{H}, fmt {fmt})"


@pytest.mark.parametrize("seed", range(_OFF, _OFF + 30))
def test_random_depth_post_processing(gpu, oracle, synth, seed):
    rng = np.random.default_rng(4000 + seed)
    rows, cols = int(rng.integers(8, 120)), int(rng.integers(8, 200))
    z = 1.0 + 8.0 * rng.random((rows, cols)) ** 2
    curr = (1000 * z + rng.normal(0, 20, z.shape)).astype(np.int16)
    prev = (1000 * z * (1 + 0.1 * rng.normal(0, 1, z.shape) * (rng.random(z.shape) < 0.3))).astype(np.int16)
    curr[rng.random(z.shape) < 0.1] = 0
    prev[rng.random(z.shape) < 0.1] = 0
    prev[rng.random(z.shape) < 0.02] = -3
    ang = rng.normal(0, 0.05)
    Tpc = synth.pose_matrix(synth.look_rotation(ang, rng.normal(0, 0.03)), rng.normal(0, 0.2, 3)).astype(np.float32)
    f = float(rng.uniform(20, 300))
    intr = (f, f * float(rng.uniform(0.9, 1.1)), float(rng.uniform(0, rows)), float(rng.uniform(0, cols)))
    thr, area = float(rng.uniform(0.0, 0.2)), float(rng.uniform(0.0, 0.9))
    g, gc = gpu.depth_post_processing(curr, prev, Tpc, intr, thr, area)
    o, oc = oracle.depth_post_processing(curr, prev, Tpc, intr, thr, area)
    assert gc == oc and np.array_equal(g, o), f"seed {seed}"


@pytest.mark.parametrize("seed", range(_OFF, _OFF + 16))
def test_random_tracker(pkg, synth, gpu, oracle, seed):
    rng = np.random.default_rng(6000 + seed)
    W, H = (160, 120) if seed % 2 else (128, 96)
    levels = int(rng.integers(1, 5))
    regime = [int(rng.choice([1, 2, 3, 3, 4])) for _ in range(levels)]
    tp = pkg.TrackerParams(levels=levels, run_till_level=int(rng.integers(0, min(2, levels))),
                           dist_thresh=float(rng.choice([0.01, 0.04, 0.0025])),
                           termination_threshold=float(rng.choice([1e-3, 1e-4, 1e-2])), regime=regime)
    d = synth.pose_matrix(synth.look_rotation(rng.normal(0, 0.01), rng.normal(0, 0.01)), rng.normal(0, 0.01, 3))
    out = {}
    for name, api in (("gpu", gpu), ("oracle", oracle)):
        wl = synth.s_room(W, H, scale=2.0)
        p = pkg.SceneParams(**wl.scene_kwargs)
        s = api.create_scene(p)
        rs, v = api.create_render_state(s, W, H), api.create_view(W, H)
        for i in range(3):
            rgba, mm, M = wl.frame(i)
            api.view_update(v, rgba, mm, timestamp=float(i))
            api.process_frame(s, v, rs, M, wl.intr)
        M2 = wl.frame(2)[2]
        api.create_icp_maps(s, rs, M2, wl.intr, download=False) if name == "gpu" else api.create_icp_maps(s, rs, M2, wl.intr)
        rgba, mm, M3 = wl.frame(3)
        api.view_update(v, rgba, mm, timestamp=3.0)
        start = (np.asarray(M2, np.float64) @ d).astype(np.float32)
        out[name] = api.track_camera(v, rs, M2, start, wl.intr, tp)
    (gp, gr), (op, orr) = out["gpu"], out["oracle"]
    assert (gr.iterations, gr.valid_points_last) == (orr.iterations, orr.valid_points_last), f"seed {seed}: control flow"
    assert np.abs(gp - op).max() <= 1e-5, f"seed {seed}: pose"


@pytest.mark.parametrize("seed", range(_OFF, _OFF + 10))
def test_random_online_correction(pkg, synth, gpu, oracle, seed):
    """Random ORB-SLAM2 keyframe sets (moved, missing, bad) through the Python OnlineCorrection scheduler with the
    keyframe store, on both engines: same selections, byte-identical maps."""
    import util
    from dslam_amd.harness import reintegrate
    rng = np.random.default_rng(7000 + seed)
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl)
    n_frames = int(rng.integers(6, 11))
    corr, start, max_age = int(rng.integers(1, 4)), int(rng.integers(1, 4)), int(rng.integers(3, 7))
    Twc = [np.linalg.inv(np.asarray(wl.frame(i)[2], np.float64)).astype(np.float32) for i in range(n_frames)]
    current = [t.copy() for t in Twc]
    sets = []
    for i in range(n_frames):
        for j in range(i):
            if rng.random() < 0.4:
                k = 1 + 5 * rng.random()
                dlt = synth.pose_matrix(synth.look_rotation(0.003 * k, -0.002 * k), [0.002 * k, 0.001 * k, -0.003 * k])
                current[j] = (np.asarray(current[j], np.float64) @ dlt).astype(np.float32)
        sets.append([(float(j), current[j].copy(), bool(rng.random() < 0.1)) for j in range(i + 1) if j == i or rng.random() < 0.85])
    res = {}
    for name, api in (("gpu", gpu), ("oracle", oracle)):
        s = api.create_scene(p)
        rs, v = api.create_render_state(s, wl.W, wl.H), api.create_view(wl.W, wl.H)
        db = reintegrate.FusionFrameDatabase(api, wl.W, wl.H, n_frames)
        log = []
        for i in range(n_frames):
            rgba, mm, _ = wl.frame(i)
            api.view_update(v, rgba, mm, timestamp=float(i))
            slot = db.insert_from_view(float(i), Twc[i], v)
            order, culled = db.online_correction(s, v, rs, wl.intr, sets[i], corr, start)
            if float(i) in db.entries:
                api.view_update_from_store(v, db.store, slot, timestamp=float(i))
                api.process_frame(s, v, rs, db.pose_to_M(Twc[i]), wl.intr)
            if len(db) > max_age:
                api.slide_window(s, rs, max_age)
                for _ in range(corr):
                    api.slide_window_defusion_part(s, rs, max_age, max(1, (max_age - start) * corr))
                db.slide_window_pose(max_age)
            log.append((order, culled, len(db)))
        res[name] = (log, util.snapshot(api, s, rs))
    assert res["gpu"][0] == res["oracle"][0], f"seed {seed}: schedules differ"
    util.assert_same_state(res["gpu"][1], res["oracle"][1], f"seed {seed}")
