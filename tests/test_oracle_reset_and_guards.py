"""CPU tests of the oracle for the round-2 parity holes: ResetScene on a used map (InfiniTamDriver.h:354-360), the
InfiniTAM_IMAGE_SCENERAYCAST picture (InfiniTamDriver.cpp:28-29), and the NaN / denormal guards of the voxel update."""
import numpy as np
import pytest

import scenarios
import util


@pytest.mark.parametrize("swapping", [0, 1])
def test_reset_scene_on_used_map_equals_fresh_scene(pkg, synth, oracle, swapping):
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl, use_swapping=swapping)
    out = scenarios.reset_scenario(oracle, pkg, wl, p)
    used = out["used"]["stats"]
    assert used["frame_counter"] == 6 + swapping and used["fusion_fifo_len"] == 2 and (out["img_used"] > 0).sum() > 500
    if not swapping:
        assert used["slid_block_count"] > 0
    if swapping:  # (blocks that leave the view are swapped out by ProcessFrame itself before the window gets to them)
        assert (out["used"]["hash"]["ptr"] == -1).sum() > 0 and out["parked_before"] > 0, "no block was parked on the host"
        assert out["stored_after_reset"] == 0
        scenarios.assert_same_full_state(out["reset_flushed"], out["fresh"], "reset + flush vs fresh scene")
    # ResetScene gives exactly a new scene: first block handed out is the top slot again
    scenarios.assert_same_full_state(out["reset"], out["fresh"], "reset vs fresh scene")
    h = out["reset"]["hash"]
    assert (h["ptr"] == -2).all() and (h["offset"] == 0).all() and not h["pos"].any()
    assert out["reset"]["stats"]["last_free_block_id"] == p.num_local_blocks - 1
    assert np.array_equal(out["reset"]["alloc_list"], np.arange(p.num_local_blocks))
    assert not (out["img_reset"] > 0).any(), "GetImage after the reset still shows the old map (stale memo?)"
    # the second life: identical map to a scene that never had the first one.  (The local map's render state outlives
    # the reset with its old visible list, exactly as upstream's does, so only the scene side is compared.)
    a, b = dict(out["reused"]), dict(out["fresh_used"])
    for k in ("visible_ids", "visible_types"):
        a.pop(k), b.pop(k)
    scenarios.assert_same_full_state(a, b, "map re-used after reset vs fresh map")
    assert (out["img_reused"] > 0).sum() > 500


def test_raycast_image_is_the_grey_shading_of_the_icp_maps(pkg, synth, oracle):
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl)
    s, rs, v = util.run_sequence(oracle, pkg, wl, p, 4)
    with pytest.raises(pkg.DslamError):
        oracle.download_raycast_image(rs)  # nothing has drawn it yet
    M = wl.frame(3)[2]
    pts, nrm = oracle.create_icp_maps(s, rs, M, wl.intr)
    img = oracle.download_raycast_image(rs)
    found = pts[..., 3] > 0
    assert found.sum() > 500 and np.array_equal(img[..., 0] != 0, found) or (img[found][:, 0] >= 51).all()
    # drawPixelGrey: every channel (uchar)((0.8 * angle + 0.2) * 255), angle = normal . (-invM column 2), in float32
    invM = np.linalg.inv(M.astype(np.float64)).astype(np.float32)
    light = -invM[:3, 2]
    n = nrm[..., :3]
    angle = (n[..., 0] * light[0] + n[..., 1] * light[1]) + n[..., 2] * light[2]
    want = np.where(found, ((np.float32(0.8) * angle + np.float32(0.2)) * np.float32(255.0)).astype(np.uint8), 0)
    assert np.abs(img[..., 0].astype(int) - want.astype(int)).max() <= 1  # (the test's float order may differ by 1 LSB)
    assert (img[..., 0] == img[..., 1]).all() and (img[..., 0] == img[..., 2]).all() and (img[..., 0] == img[..., 3]).all()
    assert not img[~found].any()


def test_degenerate_poses_never_index_with_nan(pkg, synth, oracle):
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl)
    base, after_denormal, after_nan_rgb = scenarios.degenerate_pose_scenario(oracle, pkg, wl, p)
    # (a) a denormal camera depth counts as "not in front of the camera": nothing changes
    assert np.array_equal(base.view(np.uint64), after_denormal.view(np.uint64))
    # (b) NaN colour projection: the depth channel is fused a second time, the colour channel is left alone
    assert (after_nan_rgb["w_depth"] >= base["w_depth"]).all() and (after_nan_rgb["w_depth"] > base["w_depth"]).sum() > 1000
    assert np.array_equal(after_nan_rgb["w_color"], base["w_color"]) and np.array_equal(after_nan_rgb["clr"], base["clr"])


def test_weight_params_are_validated(pkg, oracle):
    oracle.set_fusion_weight_params(True, 255, 10.0)
    for bad in ((True, 256, 10.0), (True, 0, 10.0), (True, 4, 0.0)):
        with pytest.raises(pkg.DslamError):
            oracle.set_fusion_weight_params(*bad)
    oracle.set_fusion_weight_params(False, 1, 1.0)


def test_deintegration_from_the_stored_visible_list(pkg, synth, oracle):
    """dslam_deprocess_frame_stored: fuse a keyframe into an empty map, keep its visible list, de-integrate from the list
    -> empty map again, without an allocation pass and without touching the render state."""
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl)
    s = oracle.create_scene(p)
    rs, v = oracle.create_render_state(s, wl.W, wl.H), oracle.create_view(wl.W, wl.H)
    store = oracle.create_frame_store(wl.W, wl.H, 2)
    oracle.frame_store_enable_lists(store, s)
    rgba, mm, M = wl.frame(0)
    oracle.view_update(v, rgba, mm)
    oracle.process_frame(s, v, rs, M, wl.intr)
    oracle.frame_store_put_visible_list(store, 0, s, rs)
    ids = oracle.download_visible_ids(rs)
    assert (oracle.download_voxel_blocks(s)["w_depth"] > 0).sum() > 10000
    oracle.deprocess_frame_stored(s, v, store, 0, M, wl.intr)
    vox = oracle.download_voxel_blocks(s)
    assert (vox["w_depth"] == 0).all() and (vox["sdf"] == 32767).all() and (vox["w_color"] == 0).all()
    assert np.array_equal(oracle.download_visible_ids(rs), ids)


@pytest.mark.parametrize("maintenance", [False, True])
def test_stored_list_correction_batch(pkg, synth, oracle, maintenance):
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl)
    out = scenarios.stored_list_scenario(oracle, pkg, synth, wl, p, maintenance)
    assert out["rs_untouched"] and out["missing_list_refused"]
    a, b, c = out["fused"]["voxels"], out["after_first_deintegration"]["voxels"], out["corrected"]["voxels"]
    n1 = int((a.view(np.uint64) != b.view(np.uint64)).any(axis=1).sum())
    n2 = int((a.view(np.uint64) != c.view(np.uint64)).any(axis=1).sum())
    assert n1 > 100 and n2 > 100
    # de-integration only ever lowers weights
    assert (b["w_depth"].astype(int) <= a["w_depth"].astype(int)).all()
    util.check_invariants(out["corrected"], p)
