"""GPU parity for the round-2 parity holes: ResetScene on a used map (InfiniTamDriver.h:354-360), the
InfiniTAM_IMAGE_SCENERAYCAST picture (InfiniTamDriver.cpp:28-29), the NaN / denormal guards of the voxel update and
the WeightParams bound.  HIP engine vs CPU oracle, byte for byte."""
import numpy as np
import pytest

import scenarios
import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("swapping", [0, 1])
def test_reset_scene_on_used_map(pkg, synth, gpu, oracle, swapping):
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl, use_swapping=swapping)
    g = scenarios.reset_scenario(gpu, pkg, wl, p)
    o = scenarios.reset_scenario(oracle, pkg, wl, p)
    for stage in ("used", "reset", "reused", "fresh", "fresh_used") + (("reset_flushed",) if swapping else ()):
        scenarios.assert_same_full_state(g[stage], o[stage], f"{stage} (swapping={swapping})")
    scenarios.assert_same_full_state(g["reset"], g["fresh"], "HIP: reset vs fresh scene")
    if swapping:
        assert g["parked_before"] == o["parked_before"] > 0 and g["stored_after_reset"] == 0
        scenarios.assert_same_full_state(g["reset_flushed"], g["fresh"], "HIP: reset + flush vs fresh scene")
    for k in ("img_used", "img_reset", "img_reused"):
        assert np.array_equal(g[k] > 0, o[k] > 0) and np.abs(g[k] - o[k]).max() <= 1e-4, k
    assert not (g["img_reset"] > 0).any(), "GetImage memo survived ResetScene"
    assert (g["img_used"] > 0).sum() > 500 and (g["img_reused"] > 0).sum() > 500


def test_reset_scene_full_size_default_pools(pkg, synth, gpu):
    """640x480 S-street at the default pool sizes: after a reset the 1 GiB voxel array, the tables and the rings are
    those of a new scene (sampled for the voxels: first, last and 64 used slots)."""
    wl = synth.s_street(640, 480)
    p = pkg.SceneParams(**wl.scene_kwargs)
    s = gpu.create_scene(p)
    rs, v = gpu.create_render_state(s, wl.W, wl.H), gpu.create_view(wl.W, wl.H)
    scenarios.use_map(gpu, pkg, wl, s, rs, v, range(3), slide=None, decay=None)
    used = np.nonzero(gpu.download_hash_table(s)["ptr"] >= 0)[0]
    slots = gpu.download_hash_table(s)["ptr"][used[:: max(1, len(used) // 64)]]
    assert len(used) > 5000
    gpu.reset_scene(s)
    st = gpu.stats(s)
    assert st["last_free_block_id"] == s.params.num_local_blocks - 1 and st["last_free_excess_id"] == s.params.num_excess - 1
    assert st["frame_counter"] == 0 and st["fusion_fifo_len"] == 0
    h = gpu.download_hash_table(s)
    assert (h["ptr"] == -2).all() and not h["offset"].any()
    assert np.array_equal(gpu.download_allocation_list(s), np.arange(s.params.num_local_blocks))
    assert (gpu.download_last_seen(s) == -1).all()
    for slot in [0, s.params.num_local_blocks - 1] + [int(x) for x in slots]:
        b = gpu.download_voxel_blocks(s, slot, 1)
        assert (b["sdf"] == 32767).all() and not b["w_depth"].any() and not b["clr"].any() and not b["w_color"].any()
    scenarios.use_map(gpu, pkg, wl, s, rs, v, range(1), slide=None, decay=None)
    assert gpu.download_hash_table(s)["ptr"].max() == s.params.num_local_blocks - 1


def test_raycast_image_parity(pkg, synth, gpu, oracle):
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl)
    res = {}
    for name, api in (("gpu", gpu), ("oracle", oracle)):
        s, rs, v = util.run_sequence(api, pkg, wl, p, 4)
        with pytest.raises(pkg.DslamError):
            api.download_raycast_image(rs)
        M = synth.world_to_camera(wl.pose(3) @ synth.pose_matrix(synth.look_rotation(0.04, 0.02), [0.02, -0.01, 0.01]))
        pts, nrm = api.create_icp_maps(s, rs, M, wl.intr)
        res[name] = (api.download_raycast_image(rs), pts, nrm)
    (gi, gp, gn), (oi, op, on) = res["gpu"], res["oracle"]
    assert np.array_equal(gp[..., 3], op[..., 3]) and (gp[..., 3] > 0).sum() > 500
    assert np.array_equal(gi, oi), "SCENERAYCAST image differs from the oracle"
    assert (gi[..., 0] != 0).sum() == (gp[..., 3] > 0).sum()


def test_degenerate_poses(pkg, synth, gpu, oracle):
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl)
    g = scenarios.degenerate_pose_scenario(gpu, pkg, wl, p)
    o = scenarios.degenerate_pose_scenario(oracle, pkg, wl, p)
    for a, b, what in zip(g, o, ("fused", "denormal camera depth", "NaN colour projection")):
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), what
    assert np.array_equal(g[0].view(np.uint64), g[1].view(np.uint64))
    # the same two poses through the de-integration kernel variant
    s = gpu.create_scene(p)
    rs, v = gpu.create_render_state(s, wl.W, wl.H), gpu.create_view(wl.W, wl.H)
    so = oracle.create_scene(p)
    rso, vo = oracle.create_render_state(so, wl.W, wl.H), oracle.create_view(wl.W, wl.H)
    rgba, mm, M = wl.frame(0)
    Md = np.eye(4, dtype=np.float32)
    Md[2, :] = 0.0
    Md[2, 2] = np.float32(1e-39)
    for api, sc, r, vw in ((gpu, s, rs, v), (oracle, so, rso, vo)):
        api.view_update(vw, rgba, mm)
        api.process_frame(sc, vw, r, M, wl.intr)
        api.process_frame(sc, vw, r, M, wl.intr)
        ids = api.download_visible_ids(r)
        api.deprocess_frame(sc, vw, r, M, wl.intr, M_rgb=np.zeros((4, 4), np.float32), intr_rgb=wl.intr)
        api.upload_visible_ids(r, ids)
    assert np.array_equal(gpu.download_voxel_blocks(s).view(np.uint64), oracle.download_voxel_blocks(so).view(np.uint64))


def test_weight_params_bound(pkg, synth, gpu, oracle):
    """max_new_w is bounded by the byte a voxel weight is (and by the kernel's reciprocal table): 255 works and matches
    the oracle, 256 is refused by both."""
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl, max_w=255)
    res = {}
    for name, api in (("gpu", gpu), ("oracle", oracle)):
        for bad in ((True, 256, 10.0), (True, 0, 10.0), (True, 4, 0.0)):
            with pytest.raises(pkg.DslamError):
                api.set_fusion_weight_params(*bad)
        api.set_fusion_weight_params(True, 255, 6.0)
        try:
            s, rs, v = util.run_sequence(api, pkg, wl, p, 3)
            res[name] = util.snapshot(api, s, rs)
        finally:
            api.set_fusion_weight_params(False, 1, 1.0)
    util.assert_same_state(res["gpu"], res["oracle"], "max_new_w = 255")
    assert res["gpu"]["voxels"]["w_depth"].max() == 255


def test_visible_list_replaced_between_allocation_passes(pkg, synth, gpu, oracle):
    """The allocation pass re-arms the PREVIOUS VISIBLE LIST as type 3 (upstream: `for i < noVisibleEntries:
    visibleType[visibleIDs[i]] = 3`), and the types array keeps whatever it held.  The engine encodes "previous pass" in
    a generation bit of the types instead of walking the list, so a list that was replaced behind the types' back --
    FindVisibleBlocks into the local map's own render state, an uploaded list -- has to be folded in first.  Same calls
    on both engines; list, types and map must agree after every pass."""
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl)
    objs = {}
    for name, api in (("gpu", gpu), ("oracle", oracle)):
        s = api.create_scene(p)
        objs[name] = (api, s, api.create_render_state(s, wl.W, wl.H), api.create_view(wl.W, wl.H))
    M_side = synth.world_to_camera(wl.pose(0) @ synth.pose_matrix(synth.look_rotation(0.6, 0.1), [0.1, 0.0, 0.0]))
    for i in range(8):
        rgba, mm, M = wl.frame(i)
        snaps = {}
        for name, (api, s, rs, v) in objs.items():
            api.view_update(v, rgba, mm, timestamp=float(i))
            if i == 3:    # the list becomes "everything allocated that a sideways camera sees"; the types stay
                api.find_visible_blocks(s, rs, M_side, wl.intr)
            if i == 5:    # the list becomes a thinned-out copy of itself
                api.upload_visible_ids(rs, api.download_visible_ids(rs)[::3])
            if i == 6:    # an allocation pass that only updates the list, from the sideways pose
                api.allocate_scene_from_depth(s, v, rs, M_side, wl.intr, only_update_visible_list=True)
            api.process_frame(s, v, rs, M, wl.intr)
            snaps[name] = util.snapshot(api, s, rs)
        util.assert_same_state(snaps["gpu"], snaps["oracle"], f"frame {i}")
    assert len(snaps["gpu"]["visible_ids"]) > 200


@pytest.mark.parametrize("maintenance", [False, True])
def test_stored_visible_list_deintegration_parity(pkg, synth, gpu, oracle, maintenance):
    """dslam_frame_store_put_visible_list / dslam_deprocess_frame_stored against the oracle: an OnlineCorrection-style
    batch whose de-integrations go straight to the integration kernel with the keyframe's own block list (entries that
    hold another block by now are skipped), on a plain map and on a decayed + slid one."""
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl)
    g = scenarios.stored_list_scenario(gpu, pkg, synth, wl, p, maintenance)
    o = scenarios.stored_list_scenario(oracle, pkg, synth, wl, p, maintenance)
    for stage in ("fused", "after_first_deintegration", "corrected"):
        scenarios.assert_same_full_state(g[stage], o[stage], f"{stage} (maintenance={maintenance})")
    assert g["rs_untouched"] and g["missing_list_refused"]


def test_stored_list_deintegration_equals_the_reference_call_on_an_unchanged_map(pkg, synth, gpu):
    """dslam_deprocess_frame_stored is not the reference's call sequence (the reference runs an allocation pass at the old pose
    and de-integrates what that pass lists; InfiniTamDriver.h:215-220) -- parity unpinned, like everything on this path.
    What ties it to the reference-shaped call: while the map has not changed since the keyframe was fused, the stored list IS
    the list that allocation pass produces, so both calls must leave the same voxels (round-2 ADVICE)."""
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl)
    states = []
    for stored in (False, True):
        scene = gpu.create_scene(p)
        rs, view = gpu.create_render_state(scene, wl.W, wl.H), gpu.create_view(wl.W, wl.H)
        store = gpu.create_frame_store(wl.W, wl.H, 4)
        gpu.frame_store_enable_lists(store, scene)
        for i in range(4):
            rgba, mm, M = wl.frame(i)
            gpu.view_update(view, rgba, mm, timestamp=float(i))
            gpu.frame_store_put_view(store, i, view)
            gpu.process_frame(scene, view, rs, M, wl.intr)
            gpu.frame_store_put_visible_list(store, i, scene, rs)
        # the last keyframe, de-integrated at the pose it was fused with, right after its fusion
        rgba, mm, M = wl.frame(3)
        gpu.view_update_from_store(view, store, 3, timestamp=3.0)
        if stored:
            gpu.deprocess_frame_stored(scene, view, store, 3, M, wl.intr)
        else:
            gpu.deprocess_frame(scene, view, rs, M, wl.intr)
        states.append((gpu.download_hash_table(scene), gpu.download_voxel_blocks(scene)))
        assert gpu.stats(scene, rs)["no_visible_entries"] > 50
    assert np.array_equal(states[0][0], states[1][0]), "hash table"
    assert np.array_equal(states[0][1], states[1][1]), "voxels"


@pytest.mark.parametrize("bits,what", [(1, "order key"), (2, "tile count")])
def test_device_side_errors_reach_the_caller(pkg, synth, gpu, bits, what):
    """Two conditions only a kernel can detect (an allocation ray longer than the order key encodes; a tile count of an
    ordered compaction that never arrived).  A one-thread kernel reports them exactly as a failing pass would
    (dslam_debug_inject_device_error -> report_error).  Synchronous engine: the call itself fails.  Asynchronous engine
    (what the ITMLib mirror runs): the call returns, the NEXT call that waits for the stream fails -- once --, further waits
    are clean, and dslam_get_stats keeps reporting the scene's own sticky flags until the scene is reset
    (include/dslam_fusion.h, conventions)."""
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl)
    scene = gpu.create_scene(p)
    other = gpu.create_scene(p)
    rs = gpu.create_render_state(scene, wl.W, wl.H)
    with pytest.raises(pkg.DslamError, match=what):
        gpu.debug_inject_device_error(scene, bits)        # synchronous: told by the call that caused it
    gpu.synchronize()                                      # ... and told once
    gpu.set_async(True)
    try:
        gpu.debug_inject_device_error(scene, bits)        # enqueued, returns
        with pytest.raises(pkg.DslamError, match=what):
            gpu.synchronize()                              # the first waiting call hears of it
        gpu.synchronize()
        gpu.download_hash_table(other)                     # (a read-back is a waiting call too: nothing left to tell)
        gpu.debug_inject_device_error(scene, bits)
        with pytest.raises(pkg.DslamError, match=what):
            gpu.download_hash_table(other)                 # whichever scene the waiting call is about: the word is the engine's
    finally:
        gpu.set_async(False)
    with pytest.raises(pkg.DslamError, match=what):
        gpu.stats(scene, rs)                               # sticky for the scene it happened in
    assert gpu.stats(other)["num_allocated_blocks"] == p.num_local_blocks   # ... and only there
    gpu.reset_scene(scene)
    assert gpu.stats(scene, rs)["last_free_block_id"] == p.num_local_blocks - 1


def test_device_numa_node_and_page_locked_arenas(pkg, gpu):
    """dslam_device_numa_node answers from sysfs (-1 where the platform does not say), rejects a device that does not exist;
    dslam_host_alloc hands out page-locked buffers back to back (one arena), so that a frame's two images go up as one copy,
    and frees cleanly."""
    import os
    node = gpu.device_numa_node(0)
    assert node >= -1
    if node >= 0:
        assert os.path.isdir(f"/sys/devices/system/node/node{node}")
    with pytest.raises(pkg.DslamError):
        gpu.device_numa_node(4096)
    for attempt in range(2):   # (a pair may straddle the end of an arena other tests have filled: the next pair cannot)
        a = gpu.host_alloc((480, 640, 4), np.uint8)
        b = gpu.host_alloc((480, 640), np.int16)
        if b.ctypes.data == a.ctypes.data + ((a.nbytes + 255) & ~255) or attempt == 1:
            break
    assert b.ctypes.data == a.ctypes.data + ((a.nbytes + 255) & ~255)   # the next 256-byte boundary behind `a`
    a[...] = 7
    b[...] = -3
    assert int(a.sum()) == 7 * a.size and int(b.sum()) == -3 * b.size
    big = gpu.host_alloc((16 << 20,), np.uint8)   # larger than an arena piece: an allocation of its own
    assert not (a.ctypes.data <= big.ctypes.data < a.ctypes.data + (32 << 20))
    for x in (a, b, big):
        gpu.host_free(x)
    c = gpu.host_alloc((480, 640, 4), np.uint8)   # (freed space is handed out again or returned to the system: either way usable)
    assert int(c.sum()) == 0
    gpu.host_free(c)
