"""CPU checks of the oracle's depth tracker (ITMDepthTracker::TrackCamera restated from upstream InfiniTAM v2; the
reference reaches it through trackingController->Track, InfiniTamDriver.h:151-163).  Parity unpinned: the tests pin
behaviour (it pulls a wrong pose towards the true one, it leaves a right pose alone), not reference outputs."""
import numpy as np


def _pose_err(A, B):
    D = np.linalg.inv(np.asarray(A, np.float64)) @ np.asarray(B, np.float64)
    ang = np.degrees(np.arccos(min(1.0, max(-1.0, (np.trace(D[:3, :3]) - 1.0) * 0.5))))
    return ang, float(np.linalg.norm(D[:3, 3]))


def _map_and_view(pkg, synth, api, W=160, H=120, frames=3):
    wl = synth.s_room(W, H)
    p = pkg.SceneParams(**wl.scene_kwargs)
    s = api.create_scene(p)
    rs, v = api.create_render_state(s, wl.W, wl.H), api.create_view(wl.W, wl.H)
    for i in range(frames):
        rgba, mm, M = wl.frame(i)
        api.view_update(v, rgba, mm, timestamp=float(i))
        api.process_frame(s, v, rs, M, wl.intr)
    return wl, s, rs, v


def test_tracker_pulls_a_perturbed_pose_back(pkg, synth, oracle):
    oracle.set_threads(8)
    wl, s, rs, v = _map_and_view(pkg, synth, oracle)
    _, _, M2 = wl.frame(2)
    oracle.create_icp_maps(s, rs, M2, wl.intr)
    # the view still holds frame 2: the right answer is M2
    d = synth.pose_matrix(synth.look_rotation(0.01, 0.005), [0.01, -0.005, 0.008])
    start = (np.asarray(M2, np.float64) @ d).astype(np.float32)
    pose, res = oracle.track_camera(v, rs, M2, start, wl.intr)
    a0, t0 = _pose_err(start, M2)
    a1, t1 = _pose_err(pose, M2)
    assert res.iterations >= 3 and res.valid_points_last > 1000
    assert a1 < 0.6 * a0 and t1 < 0.6 * t0
    R = pose[:3, :3].astype(np.float64)
    assert np.abs(R @ R.T - np.eye(3)).max() < 1e-6 and np.array_equal(pose[3], [0, 0, 0, 1])  # Coerce: a rigid pose
    # started at the right pose, a single-level tracker stays there (to within the noise of the 5 mm voxel map) ...
    pose2, _ = oracle.track_camera(v, rs, M2, M2, wl.intr, pkg.TrackerParams(levels=1, regime=[3]))
    a2, t2 = _pose_err(pose2, M2)
    assert a2 < 0.03 and t2 < 0.0005
    # ... while the 5-level default drifts by the pyramid's quarter-pixel convention (intrinsics * 0.5 per level, as
    # upstream) that its two full-resolution iterations do not fully undo: 0.28 deg at 160x120, 0.15 deg at 320x240
    pose3, _ = oracle.track_camera(v, rs, M2, M2, wl.intr)
    a3, t3 = _pose_err(pose3, M2)
    assert a3 < 0.4 and t3 < 0.008


def test_tracker_follows_the_next_frame(pkg, synth, oracle):
    wl, s, rs, v = _map_and_view(pkg, synth, oracle)
    _, _, M2 = wl.frame(2)
    rgba, mm, M3 = wl.frame(3)
    oracle.create_icp_maps(s, rs, M2, wl.intr)
    oracle.view_update(v, rgba, mm, timestamp=3.0)
    pose, res = oracle.track_camera(v, rs, M2, M2, wl.intr)  # DenseSlam.cpp:200-206: start from the previous pose
    a0, t0 = _pose_err(M2, M3)
    a1, t1 = _pose_err(pose, M3)
    assert a1 < 0.75 * a0 and t1 < 0.75 * t0


def test_tracker_regimes_and_degenerate_input(pkg, synth, oracle):
    wl, s, rs, v = _map_and_view(pkg, synth, oracle)
    _, _, M2 = wl.frame(2)
    oracle.create_icp_maps(s, rs, M2, wl.intr)
    d = synth.pose_matrix(synth.look_rotation(0.0, 0.0), [0.012, 0.0, 0.0])
    start = (np.asarray(M2, np.float64) @ d).astype(np.float32)
    # translation-only regime must not touch the rotation
    pose, _ = oracle.track_camera(v, rs, M2, start, wl.intr, pkg.TrackerParams(levels=2, regime=[2, 2]))
    assert np.abs(pose[:3, :3] - start[:3, :3]).max() < 1e-6
    assert _pose_err(pose, M2)[1] < _pose_err(start, M2)[1]
    # NONE on every level: pose returned unchanged, no evaluation
    pose, res = oracle.track_camera(v, rs, M2, start, wl.intr, pkg.TrackerParams(levels=3, regime=[4, 4, 4]))
    assert res.iterations == 0 and np.array_equal(pose, start)
    # an empty depth image: every evaluation is rejected, the pose comes back unchanged
    rgba, mm, _ = wl.frame(2)
    oracle.view_update(v, rgba, np.zeros_like(mm))
    pose, res = oracle.track_camera(v, rs, M2, start, wl.intr)
    assert res.valid_points_last == 0 and np.abs(pose - start).max() < 1e-6
