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


def test_batch_of_32_keyframes_at_bench_size_equals_the_loop(pkg, synth, gpu):
    """The bench's batch: 32 keyframes of a 48-keyframe S-street map (640x480, 0x40000-block pool, decay + window before the
    batch), every one of the 64 operation bits in use, ~30 k touched blocks.  The oracle would need a minute for this; the
    HIP engine's own per-keyframe loop (checked against the oracle at the sizes above) defines the expected bytes: map,
    rings, free lists, render state, and the stored lists (a second batch back to the old poses de-integrates from the lists
    the first one left)."""
    wl = synth.s_street(640, 480)
    p = pkg.SceneParams(num_local_blocks=0x40000, num_buckets=0x100000, num_excess=0x20000, history_words=4, **wl.scene_kwargs)
    n_map, K = 48, 32
    frames = [wl.frame(i) for i in range(n_map)]
    ids = list(range(n_map - K, n_map))
    new = [synth.world_to_camera(wl.pose(i) @ synth.pose_matrix(synth.look_rotation(0.002 * (k + 1), 0.0), [0.01 * (k + 1), 0.0, 0.02])) for k, i in enumerate(ids)]
    old = [frames[i][2] for i in ids]
    states = {}
    for call in ("loop", "batch"):
        scene = gpu.create_scene(p)
        rs, view = gpu.create_render_state(scene, wl.W, wl.H), gpu.create_view(wl.W, wl.H)
        store = gpu.create_frame_store(wl.W, wl.H, n_map)
        gpu.frame_store_enable_lists(store, scene)
        for i, (rgba, mm, M) in enumerate(frames):
            gpu.view_update(view, rgba, mm, timestamp=float(i))
            gpu.frame_store_put_view(store, i, view)
            gpu.process_frame(scene, view, rs, M, wl.intr)
            gpu.frame_store_put_visible_list(store, i, scene, rs)
            if i + 1 > 40:
                gpu.slide_window(scene, rs, 40)
            gpu.decay(scene, rs, 3, 30, True)
        out = []
        for a, b in ((old, new), (new, old)):
            if call == "batch":
                gpu.reintegrate_batch(scene, view, rs, store, ids, a, b, wl.intr)
            else:
                for k, i in enumerate(ids):
                    gpu.view_update_from_store(view, store, i, timestamp=float(i))
                    gpu.deprocess_frame_stored(scene, view, store, i, a[k], wl.intr)
                    gpu.process_frame(scene, view, rs, b[k], wl.intr, is_defusion=True)
                    gpu.frame_store_put_visible_list(store, i, scene, rs)
            out.append(scenarios.full_state(gpu, scene, rs))
        states[call] = out
        assert gpu.stats(scene, rs)["no_visible_entries"] > 4000
    for stage in range(2):
        scenarios.assert_same_full_state(states["batch"][stage], states["loop"][stage], f"batch {stage}: one call vs the per-keyframe loop")
    assert not np.array_equal(states["batch"][0]["voxels"], states["batch"][1]["voxels"]), "the second batch must have moved the map again"


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_batch_with_block_exchange_equals_the_unsharded_loop(pkg, synth, gpu, world):
    """The SHARDED form of the block kernel (k_reintegrate_blocks<., 1>: half-block units, ring and dirty writes in front of
    the shard test) followed by the dirty-block exchange, rehearsed on one GPU: `world` replicas of one map play the ranks,
    each runs dslam_reintegrate_batch under dslam_scene_set_shard(r, world), then plan -> pack -> (what the all-gather
    delivers) -> unpack.  Every replica -- map, rings, free lists, render state, stored lists -- must equal the unsharded
    per-keyframe loop over stored lists (itself tied to the oracle by the tests above); a second batch runs from the lists
    the first one left.  Single-GPU emulation: multi-rank hardware is not available to the build."""
    torch = pytest.importorskip("torch")
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl, num_local_blocks=0x800, num_buckets=0x1000, num_excess=0x400, history_words=1)
    chunk, n_map = 16, 12
    ids = [9, 10, 11, 4, 7]
    frames = [wl.frame(i) for i in range(n_map)]
    old = [frames[i][2] for i in ids]
    new = [synth.world_to_camera(wl.pose(i) @ synth.pose_matrix(synth.look_rotation(0.004, 0.002 * (k + 1)), [0.004, -0.002, 0.003 * (k + 1)])) for k, i in enumerate(ids)]

    def build():
        scene = gpu.create_scene(p)
        rs, view = gpu.create_render_state(scene, wl.W, wl.H), gpu.create_view(wl.W, wl.H)
        store = gpu.create_frame_store(wl.W, wl.H, n_map)
        gpu.frame_store_enable_lists(store, scene)
        for i, (rgba, mm, M) in enumerate(frames):
            gpu.view_update(view, rgba, mm, timestamp=float(i))
            gpu.frame_store_put_view(store, i, view)
            gpu.process_frame(scene, view, rs, M, wl.intr)
            gpu.frame_store_put_visible_list(store, i, scene, rs)
            if i + 1 > 8:
                gpu.slide_window(scene, rs, 8)
            gpu.decay(scene, rs, 1, 2, True)
        return scene, rs, view, store

    # the definition: unsharded, keyframe by keyframe, from the stored lists
    scene, rs, view, store = build()
    want = []
    for a, b in ((old, new), (new, old)):
        for k, i in enumerate(ids):
            gpu.view_update_from_store(view, store, i, timestamp=float(i))
            gpu.deprocess_frame_stored(scene, view, store, i, a[k], wl.intr)
            gpu.process_frame(scene, view, rs, b[k], wl.intr, is_defusion=True)
            gpu.frame_store_put_visible_list(store, i, scene, rs)
        want.append(scenarios.full_state(gpu, scene, rs))

    ranks = [build() for _ in range(world)]
    for stage, (a, b) in enumerate(((old, new), (new, old))):
        counts = None
        for r, (scene, rs, view, store) in enumerate(ranks):
            gpu.track_dirty(scene, True)
            gpu.set_shard(scene, r, world, chunk)
            gpu.reintegrate_batch(scene, view, rs, store, ids, a, b, wl.intr)
            c = gpu.shard_dirty_plan(scene, world, chunk)
            assert counts is None or c == counts, "ranks disagree about the dirty lists"
            counts = c
        assert min(counts) > 10
        cap = max(counts)
        recv = torch.zeros((world, cap, 4096), dtype=torch.uint8, device="cuda")
        for r, (scene, rs, view, store) in enumerate(ranks):
            gpu.shard_dirty_pack(scene, r, recv[r].data_ptr(), cap)
        gpu.synchronize()
        for r, (scene, rs, view, store) in enumerate(ranks):
            if r == 0:   # before the exchange a rank holds only its own shard's updates
                assert not np.array_equal(gpu.download_voxel_blocks(scene).view(np.uint64), want[stage]["voxels"].view(np.uint64))
            gpu.shard_dirty_unpack(scene, r, recv.data_ptr(), cap)
            gpu.set_shard(scene, 0, 1, chunk)
            gpu.track_dirty(scene, False)
            scenarios.assert_same_full_state(scenarios.full_state(gpu, scene, rs), want[stage], f"batch {stage}, rank {r} of {world} after the exchange")
