"""dslam_reintegrate_batch (block-major: every touched voxel block loaded once) against the per-keyframe loop that defines
it (reference DenseSlam.cpp:389-403: DeProcessFrame at the old pose, ProcessFrame at the new one) -- on the HIP engine
and on the CPU oracle.  Integer / byte state: bit-exact, no tolerance."""
import numpy as np
import pytest

import scenarios
import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("maintenance", [False, True])
def test_batch_equals_loop_and_oracle(pkg, synth, gpu, oracle, maintenance):
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl, history_words=1)
    g_batch = scenarios.batch_scenario(gpu, pkg, synth, wl, p, maintenance, "batch")
    g_loop = scenarios.batch_scenario(gpu, pkg, synth, wl, p, maintenance, "loop")
    o_batch = scenarios.batch_scenario(oracle, pkg, synth, wl, p, maintenance, "batch")
    for key in ("first", "second"):
        scenarios.assert_same_full_state(g_batch[key], g_loop[key], f"{key}: HIP batch vs HIP per-keyframe loop")
        scenarios.assert_same_full_state(g_batch[key], o_batch[key], f"{key}: HIP batch vs oracle")
    (gt, gc), (ot, oc) = g_batch["alloc_scratch"], o_batch["alloc_scratch"]
    assert np.array_equal(gt, ot), "allocType of the last pass"
    assert np.array_equal(gc[ot > 0], oc[ot > 0]), "blockCoords of the last pass"
    # the batch did something: voxels changed between the two stages
    assert not np.array_equal(g_batch["first"]["voxels"].view(np.uint64), g_batch["second"]["voxels"].view(np.uint64))


def test_batch_longer_than_one_mask(pkg, synth, gpu, oracle):
    """More keyframes than one 64-bit operation mask holds (32): the batch is cut, the result is not."""
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl, history_words=2)
    picks = tuple(range(39, 1, -1))   # 38 keyframes, newest first
    g = scenarios.batch_scenario(gpu, pkg, synth, wl, p, False, "batch", n_frames=40, picks=picks, second=(5, 20, 33))
    o = scenarios.batch_scenario(oracle, pkg, synth, wl, p, False, "batch", n_frames=40, picks=picks, second=(5, 20, 33))
    for key in ("first", "second"):
        scenarios.assert_same_full_state(g[key], o[key], key)


def test_batch_full_size(pkg, synth, gpu, oracle):
    """640x480, default pools: 6 keyframes of an S-street map corrected in one batch, compared with the oracle's loop."""
    wl = synth.s_street(640, 480)
    p = pkg.SceneParams(**wl.scene_kwargs)
    g = scenarios.batch_scenario(gpu, pkg, synth, wl, p, False, "batch", n_frames=8, picks=(6, 7, 3, 5), second=(2, 6))
    o = scenarios.batch_scenario(oracle, pkg, synth, wl, p, False, "batch", n_frames=8, picks=(6, 7, 3, 5), second=(2, 6))
    for key in ("first", "second"):
        scenarios.assert_same_full_state(g[key], o[key], key)


def test_batch_refuses_what_it_cannot_do(pkg, synth, gpu):
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl, use_swapping=1)
    scene = gpu.create_scene(p)
    rs, view = gpu.create_render_state(scene, wl.W, wl.H), gpu.create_view(wl.W, wl.H)
    store = gpu.create_frame_store(wl.W, wl.H, 2)
    gpu.frame_store_enable_lists(store, scene)
    rgba, mm, M = wl.frame(0)
    gpu.view_update(view, rgba, mm)
    gpu.frame_store_put_view(store, 0, view)
    gpu.process_frame(scene, view, rs, M, wl.intr)
    gpu.frame_store_put_visible_list(store, 0, scene, rs)
    with pytest.raises(pkg.DslamError):
        gpu.reintegrate_batch(scene, view, rs, store, [0], [M], [M], wl.intr)   # a swapping scene
    with pytest.raises(pkg.DslamError):
        gpu.reintegrate_batch(scene, view, rs, store, [1], [M], [M], wl.intr)   # a keyframe without a stored list
