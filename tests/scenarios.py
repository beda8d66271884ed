"""Call sequences shared by the CPU (oracle) and GPU parity tests: each takes an `api` (the HIP engine or the oracle,
same wrapper class) and returns everything a test compares."""
import numpy as np

import util


def full_state(api, scene, rs=None):
    """util.snapshot plus the per-slot / per-entry side arrays the maintenance paths keep."""
    snap = util.snapshot(api, scene, rs)
    snap["last_seen"] = api.download_last_seen(scene)
    if scene.params.use_swapping:
        snap["swap_states"] = api.download_swap_states(scene)
    return snap


def assert_same_full_state(a, b, what=""):
    util.assert_same_state(a, b, what)
    assert np.array_equal(a["last_seen"], b["last_seen"]), f"{what}: last_seen differs"
    if "swap_states" in a:
        assert np.array_equal(a["swap_states"], b["swap_states"]), f"{what}: swap states differ"


def use_map(api, pkg, wl, scene, rs, view, frames, slide=2, decay=(1, 2, True), flush_at=None):
    """DenseSlam::ProcessFrame steps 10-13 (reference DenseSlam.cpp:210-232): fuse, slide the window, decay; with
    `flush_at`, DenseSlam::saveLocalMapToHostMemory (DenseSlam.h:248-251) after that frame."""
    for i in frames:
        rgba, mm, M = wl.frame(i)
        api.view_update(view, rgba, mm, timestamp=float(i))
        api.process_frame(scene, view, rs, M, wl.intr)
        if slide is not None and api.stats(scene, rs)["fusion_fifo_len"] > slide:
            api.slide_window(scene, rs, slide)
        if decay is not None:
            api.decay(scene, rs, *decay)
        if flush_at is not None and i == flush_at:
            api.save_to_global_memory(scene)


def reset_scenario(api, pkg, wl, params):
    """InfiniTamDriver::ResetLocalMap (InfiniTamDriver.h:354-360) on a map that has been USED: window + decay active,
    (with swapping) blocks parked on the host, a GetImage memo live; then the map is used again.  Returns the states
    and images along the way."""
    out = {}
    scene = api.create_scene(params)
    rs = api.create_render_state(scene, wl.W, wl.H)
    free = api.create_render_state(scene, wl.W, wl.H)
    view = api.create_view(wl.W, wl.H)
    use_map(api, pkg, wl, scene, rs, view, range(6))
    if params.use_swapping:  # park everything on the host, then look somewhere else: what is not seen again stays parked
        api.save_to_global_memory(scene)
        use_map(api, pkg, wl, scene, rs, view, [14])
        M5 = wl.frame(14)[2]
    else:
        M5 = wl.frame(5)[2]
    out["img_used"] = api.get_image(scene, free, M5, wl.intr, pkg.IMAGE_DEPTH)  # leaves a memo of (map version, pose)
    out["used"] = full_state(api, scene, rs)
    if params.use_swapping:
        out["parked_before"] = int(sum(api.download_stored_block(scene, int(t))[0] for t in np.nonzero(out["used"]["hash"]["ptr"] == -1)[0][:64]))
    api.reset_scene(scene)
    out["reset"] = full_state(api, scene)  # the scene alone: ResetScene does not touch the render state
    if params.use_swapping:
        out["stored_after_reset"] = int(sum(api.download_stored_block(scene, int(t))[0] for t in range(0, scene.n_entries, max(1, scene.n_entries // 257))))
    # the memo must be gone: same render state, same pose, but the map is empty now
    out["img_reset"] = api.get_image(scene, free, M5, wl.intr, pkg.IMAGE_DEPTH)
    if params.use_swapping:
        api.save_to_global_memory(scene)  # reset-then-flush: nothing resident, nothing stored
        out["reset_flushed"] = full_state(api, scene)
    use_map(api, pkg, wl, scene, rs, view, range(2, 8))
    out["reused"] = full_state(api, scene, rs)
    M7 = wl.frame(7)[2]
    out["img_reused"] = api.get_image(scene, free, M7, wl.intr, pkg.IMAGE_DEPTH)
    # a scene that never saw the first life, same second life (its own fresh render state)
    scene2 = api.create_scene(params)
    out["fresh"] = full_state(api, scene2)
    rs2 = api.create_render_state(scene2, wl.W, wl.H)
    view2 = api.create_view(wl.W, wl.H)
    use_map(api, pkg, wl, scene2, rs2, view2, range(2, 8))
    out["fresh_used"] = full_state(api, scene2, rs2)
    return out


def degenerate_pose_scenario(api, pkg, wl, params):
    """Poses no tracker produces, which must not index an image with (int)NaN: (a) a depth matrix whose third row is
    denormal (camera depth of every voxel is a denormal float), (b) a colour-camera matrix of zeros (0 / 0 = NaN
    projection).  Returns the voxel arrays after each."""
    scene = api.create_scene(params)
    rs = api.create_render_state(scene, wl.W, wl.H)
    view = api.create_view(wl.W, wl.H)
    rgba, mm, M = wl.frame(0)
    api.view_update(view, rgba, mm)
    api.process_frame(scene, view, rs, M, wl.intr)
    base = api.download_voxel_blocks(scene)
    Md = np.eye(4, dtype=np.float32)
    Md[2, :] = 0.0
    Md[2, 2] = np.float32(1e-39)  # pc.z = 1e-39 * z: denormal and > 0 for every voxel in front of the origin plane
    api.integrate_into_scene(scene, view, rs, Md, wl.intr)
    after_denormal = api.download_voxel_blocks(scene)
    api.integrate_into_scene(scene, view, rs, M, wl.intr, M_rgb=np.zeros((4, 4), np.float32), intr_rgb=wl.intr)
    after_nan_rgb = api.download_voxel_blocks(scene)
    return base, after_denormal, after_nan_rgb


def stored_list_scenario(api, pkg, synth, wl, params, maintenance):
    """DenseSlam's keyframe database with each keyframe's fusion-time visible list kept next to its images
    (dslam_frame_store_put_visible_list), then an OnlineCorrection-style batch (reference DenseSlam.cpp:390-403) whose
    de-integrations use the stored lists (dslam_deprocess_frame_stored).  Returns the states along the way."""
    n_frames = 10
    scene = api.create_scene(params)
    rs, view = api.create_render_state(scene, wl.W, wl.H), api.create_view(wl.W, wl.H)
    store = api.create_frame_store(wl.W, wl.H, n_frames)
    api.frame_store_enable_lists(store, scene)
    poses = {}
    out = {}
    for i in range(n_frames):
        rgba, mm, M = wl.frame(i)
        api.view_update(view, rgba, mm, timestamp=float(i))
        api.frame_store_put_view(store, i, view)
        api.process_frame(scene, view, rs, M, wl.intr)
        api.frame_store_put_visible_list(store, i, scene, rs)
        poses[i] = M
        if maintenance:
            if api.stats(scene, rs)["fusion_fifo_len"] > 4:
                api.slide_window(scene, rs, 4)
            api.decay(scene, rs, 1, 2, True)
    out["fused"] = full_state(api, scene, rs)
    rs_before = (api.download_visible_ids(rs), api.download_visible_types(rs))
    for n, i in enumerate((7, 9, 8, 2)):  # (keyframe 2 has left the window when `maintenance`: most of its blocks are gone)
        new_M = synth.world_to_camera(wl.pose(i) @ synth.pose_matrix(synth.look_rotation(0.003 * (n + 1), -0.002), [0.004, 0.001 * n, -0.003]))
        api.view_update_from_store(view, store, i, timestamp=float(i))
        api.deprocess_frame_stored(scene, view, store, i, poses[i], wl.intr)
        if n == 0:  # the stored-list de-integration leaves the render state alone
            out["rs_untouched"] = (np.array_equal(api.download_visible_ids(rs), rs_before[0]) and
                                   np.array_equal(api.download_visible_types(rs), rs_before[1]))
            out["after_first_deintegration"] = full_state(api, scene, rs)
        api.process_frame(scene, view, rs, new_M, wl.intr, is_defusion=True)
        api.frame_store_put_visible_list(store, i, scene, rs)  # the keyframe now lives in the blocks of its re-fusion
        poses[i] = new_M
    out["corrected"] = full_state(api, scene, rs)
    # a keyframe that was never given a list cannot be de-integrated this way
    store2 = api.create_frame_store(wl.W, wl.H, 2)
    api.frame_store_enable_lists(store2, scene)
    api.frame_store_put_view(store2, 0, view)
    try:
        api.deprocess_frame_stored(scene, view, store2, 0, poses[0], wl.intr)
        out["missing_list_refused"] = False
    except pkg.DslamError:
        out["missing_list_refused"] = True
    return out


def batch_scenario(api, pkg, synth, wl, params, maintenance, call, n_frames=12, picks=(7, 9, 8, 2, 11, 5), second=(3, 9, 7)):
    """DenseSlam::OnlineCorrection's loop (reference DenseSlam.cpp:389-403) over keyframes kept in a store with their
    fusion-time visible lists, twice (the second batch de-integrates from the lists the first one left), with a fusion in
    between.  call = "batch": dslam_reintegrate_batch; "loop": the per-keyframe calls that define it."""
    scene = api.create_scene(params)
    rs, view = api.create_render_state(scene, wl.W, wl.H), api.create_view(wl.W, wl.H)
    store = api.create_frame_store(wl.W, wl.H, n_frames + 1)
    api.frame_store_enable_lists(store, scene)
    poses = {}
    out = {}
    for i in range(n_frames):
        rgba, mm, M = wl.frame(i)
        api.view_update(view, rgba, mm, timestamp=float(i))
        api.frame_store_put_view(store, i, view)
        api.process_frame(scene, view, rs, M, wl.intr)
        api.frame_store_put_visible_list(store, i, scene, rs)
        poses[i] = M
        if maintenance:
            if api.stats(scene, rs)["fusion_fifo_len"] > 5:
                api.slide_window(scene, rs, 5)
            api.decay(scene, rs, 1, 3, True)

    def correct(ids, seed):
        new = []
        for n, i in enumerate(ids):
            new.append(synth.world_to_camera(wl.pose(i) @ synth.pose_matrix(
                synth.look_rotation(0.003 * (n + 1 + seed), -0.002 * (1 + seed)), [0.004 + 0.002 * seed, 0.001 * n, -0.003])))
        if call == "batch":
            api.reintegrate_batch(scene, view, rs, store, [], [], [], wl.intr)   # (the set-up call: an empty batch changes nothing)
            api.reintegrate_batch(scene, view, rs, store, list(ids), [poses[i] for i in ids], new, wl.intr)
        else:
            for i, new_M in zip(ids, new):
                api.view_update_from_store(view, store, i, timestamp=float(i))
                api.deprocess_frame_stored(scene, view, store, i, poses[i], wl.intr)
                api.process_frame(scene, view, rs, new_M, wl.intr, is_defusion=True)
                api.frame_store_put_visible_list(store, i, scene, rs)
        for i, new_M in zip(ids, new):
            poses[i] = new_M

    correct([i for i in picks if i < n_frames], 0)
    out["first"] = full_state(api, scene, rs)
    rgba, mm, M = wl.frame(n_frames)     # the map is used on: one more keyframe
    api.view_update(view, rgba, mm, timestamp=float(n_frames))
    api.frame_store_put_view(store, n_frames, view)
    api.process_frame(scene, view, rs, M, wl.intr)
    api.frame_store_put_visible_list(store, n_frames, scene, rs)
    poses[n_frames] = M
    correct([i for i in second if i < n_frames] + [n_frames], 1)
    out["second"] = full_state(api, scene, rs)
    out["alloc_scratch"] = api.download_alloc_scratch(scene)
    return out
