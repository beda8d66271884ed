"""Depth tracker (ICP) on the device vs the CPU oracle, through the C ABI (dslam_track_camera; reference call site
trackingController->Track, InfiniTamDriver.h:151-163).  The per-pixel terms are the same float operations on both
sides and both accumulate them in double, so only the summation order differs: the tracked pose must agree to 1e-6
per matrix entry (floating point, tolerance stated) and the control flow (iterations, valid points) exactly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(pkg, synth, api, W, H, frames=3):
    wl = synth.s_room(W, H)
    p = pkg.SceneParams(**wl.scene_kwargs)
    s = api.create_scene(p)
    rs, v = api.create_render_state(s, wl.W, wl.H), api.create_view(wl.W, wl.H)
    for i in range(frames):
        rgba, mm, M = wl.frame(i)
        api.view_update(v, rgba, mm, timestamp=float(i))
        api.process_frame(s, v, rs, M, wl.intr)
    return wl, s, rs, v


@pytest.mark.parametrize("size", [(160, 120), (640, 480)])
def test_track_camera_matches_oracle(pkg, synth, gpu, oracle, size):
    oracle.set_threads(8)
    out = {}
    for name, api in (("gpu", gpu), ("oracle", oracle)):
        wl, s, rs, v = _setup(pkg, synth, api, *size)
        _, _, M2 = wl.frame(2)
        rgba, mm, M3 = wl.frame(3)
        api.create_icp_maps(s, rs, M2, wl.intr)
        api.view_update(v, rgba, mm, timestamp=3.0)
        runs = []
        runs.append(api.track_camera(v, rs, M2, M2, wl.intr))                      # next frame from the previous pose
        d = synth.pose_matrix(synth.look_rotation(0.008, -0.004), [0.01, 0.004, -0.006])
        start = (np.asarray(M3, np.float64) @ d).astype(np.float32)
        runs.append(api.track_camera(v, rs, M2, start, wl.intr))                   # a perturbed start
        runs.append(api.track_camera(v, rs, M2, start, wl.intr, pkg.TrackerParams(levels=3, regime=[3, 2, 1])))
        out[name] = runs
    for (gp, gr), (op, orr) in zip(out["gpu"], out["oracle"]):
        assert (gr.iterations, gr.valid_points_last) == (orr.iterations, orr.valid_points_last)
        assert abs(gr.f_last - orr.f_last) <= 1e-6 * max(1.0, abs(orr.f_last))
        assert np.abs(gp - op).max() <= 1e-6
    assert out["gpu"][0][1].iterations >= 3 and out["gpu"][0][1].valid_points_last > 1000


def test_track_camera_is_deterministic_and_lazy_depth_safe(pkg, synth, gpu):
    wl, s, rs, v = _setup(pkg, synth, gpu, 160, 120)
    _, _, M2 = wl.frame(2)
    rgba, mm, _ = wl.frame(3)
    gpu.create_icp_maps(s, rs, M2, wl.intr)
    gpu.view_update(v, rgba, mm, timestamp=3.0)  # depth conversion still pending: the tracker must trigger it
    a, _ = gpu.track_camera(v, rs, M2, M2, wl.intr)
    b, _ = gpu.track_camera(v, rs, M2, M2, wl.intr)
    assert np.array_equal(a, b)
    with pytest.raises(Exception):
        fresh = gpu.create_render_state(s, wl.W, wl.H)
        gpu.track_camera(v, fresh, M2, M2, wl.intr)  # no ICP maps yet


def test_track_fuse_loop_without_ground_truth_poses(pkg, synth, gpu, oracle):
    """DenseSlam::ProcessFrame without ORB-SLAM2 odometry (DenseSlam.cpp:200-232): Prepare -> UpdateView -> Track ->
    Integrate, every pose after the first estimated by the tracker from the map built so far.  The HIP engine and the
    oracle must follow the same trajectory (the tracker's sums differ in summation order, so poses agree to 1e-4 over
    the sequence, not bit for bit) and stay near the true one.  The synthetic camera moves 1 degree and 2.6 cm per frame,
    several times a hand-held 30 Hz sensor: upstream's default regime (rotation-only coarse levels, 2 + 4 full
    iterations) follows only part of that, three full levels follow all of it -- both are run."""
    oracle.set_threads(8)
    W, H, n = 320, 240, 8
    tp = pkg.TrackerParams(levels=3, regime=[3, 3, 3], termination_threshold=1e-4)
    traj = {}
    for name, api in (("gpu", gpu), ("oracle", oracle)):
        wl = synth.s_room(W, H, scale=2.0)
        p = pkg.SceneParams(**wl.scene_kwargs)
        s = api.create_scene(p)
        rs, v = api.create_render_state(s, W, H), api.create_view(W, H)
        rgba, mm, M = wl.frame(0)
        api.view_update(v, rgba, mm, timestamp=0.0)
        api.process_frame(s, v, rs, M, wl.intr)
        poses = [np.asarray(M, np.float32)]
        for i in range(1, n):
            api.create_icp_maps(s, rs, poses[-1], wl.intr)            # PrepareNextStepLocalMap
            rgba, mm, _ = wl.frame(i)
            api.view_update(v, rgba, mm, timestamp=float(i))          # UpdateView
            est, res = api.track_camera(v, rs, poses[-1], poses[-1], wl.intr, tp)  # TrackLocalMap
            assert res.valid_points_last > 1000
            api.process_frame(s, v, rs, est, wl.intr)                 # IntegrateLocalMap
            poses.append(est)
        traj[name] = (poses, [np.asarray(wl.frame(i)[2], np.float64) for i in range(n)])
    g, o = traj["gpu"][0], traj["oracle"][0]
    assert max(np.abs(a - b).max() for a, b in zip(g, o)) <= 1e-4
    truth = traj["gpu"][1]
    D = np.linalg.inv(np.asarray(g[-1], np.float64)) @ truth[-1]
    ang = np.degrees(np.arccos(min(1.0, (np.trace(D[:3, :3]) - 1.0) * 0.5)))
    # 7 tracked frames, 7 degrees and 18 cm in total: the estimate ends within half a degree and a centimetre
    assert ang < 0.5 and np.linalg.norm(D[:3, 3]) < 0.01
