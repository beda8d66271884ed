"""Edge cases and full-size, oracle-free properties of the HIP engine (called through the C ABI)."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


def _fuse(api, pkg, wl, p, n, M_rgb_of=None, intr_rgb=None, view_size=None):
    s = api.create_scene(p)
    rs = api.create_render_state(s, wl.W, wl.H)
    v = api.create_view(wl.W, wl.H)
    for i in range(n):
        rgba, mm, M = wl.frame(i)
        api.view_update(v, rgba, mm, timestamp=float(i))
        api.process_frame(s, v, rs, M, wl.intr, M_rgb=None if M_rgb_of is None else M_rgb_of(M), intr_rgb=intr_rgb)
    return s, rs, v


def test_separate_rgb_camera_variant(pkg, synth, gpu, oracle):
    """trafo_rgb_to_depth != identity and different RGB intrinsics: exercises the two-matrix kernel variant
    (the reference sets identity, InfiniTamDriver.cpp:74-75, but the ITMLib interface allows any calib)."""
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl)
    T = synth.pose_matrix(synth.look_rotation(0.01, 0.005), [0.02, -0.01, 0.0]).astype(np.float32)
    intr_rgb = np.asarray(wl.intr, np.float32) * np.float32(1.02)
    snaps = {}
    for name, api in (("gpu", gpu), ("oracle", oracle)):
        s, rs, v = _fuse(api, pkg, wl, p, 4, M_rgb_of=lambda M: (T @ M).astype(np.float32), intr_rgb=intr_rgb)
        snaps[name] = util.snapshot(api, s, rs)
    util.assert_same_state(snaps["gpu"], snaps["oracle"], "separate RGB camera")
    assert (snaps["gpu"]["voxels"]["w_color"] > 0).sum() > 1000


def test_stop_integrating_at_max_w(pkg, synth, gpu, oracle):
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl, max_w=2, stop_integrating_at_max_w=1)
    snaps = {}
    for name, api in (("gpu", gpu), ("oracle", oracle)):
        s, rs, v = _fuse(api, pkg, wl, p, 5)
        snaps[name] = util.snapshot(api, s, rs)
    util.assert_same_state(snaps["gpu"], snaps["oracle"], "stopIntegratingAtMaxW")
    assert snaps["gpu"]["voxels"]["w_depth"].max() == 2


def test_empty_and_invalid_depth(pkg, synth, gpu, oracle):
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl)
    rgba, mm, M = wl.frame(0)
    for depth in (np.zeros_like(mm), np.full_like(mm, -5), np.full_like(mm, 32500)):  # none, negative, > 32000 mm
        snaps = {}
        for name, api in (("gpu", gpu), ("oracle", oracle)):
            s = api.create_scene(p)
            rs = api.create_render_state(s, wl.W, wl.H)
            v = api.create_view(wl.W, wl.H)
            api.view_update(v, rgba, depth)
            api.process_frame(s, v, rs, M, wl.intr)
            snaps[name] = util.snapshot(api, s, rs)
            snaps[name]["img"] = api.get_image(s, rs, M, wl.intr, pkg.IMAGE_DEPTH)
        util.assert_same_state(snaps["gpu"], snaps["oracle"], "empty depth")
        assert snaps["gpu"]["stats"]["no_visible_entries"] == 0 and (snaps["gpu"]["img"] == 0).all()


def test_shards_partition_the_integration(pkg, synth, gpu):
    """integrate(shard 0) then integrate(shard 1) over the same visible list == unsharded integrate."""
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl)
    rgba, mm, M = wl.frame(0)
    res = []
    for shards in (1, 2, 3):
        s = gpu.create_scene(p)
        rs = gpu.create_render_state(s, wl.W, wl.H)
        v = gpu.create_view(wl.W, wl.H)
        gpu.view_update(v, rgba, mm)
        gpu.allocate_scene_from_depth(s, v, rs, M, wl.intr)
        for k in range(shards):
            gpu.set_shard(s, k, shards, 8)
            gpu.integrate_into_scene(s, v, rs, M, wl.intr)
        gpu.set_shard(s, 0, 1, 8)
        res.append(gpu.download_voxel_blocks(s).view(np.uint64))
    assert np.array_equal(res[0], res[1]) and np.array_equal(res[0], res[2])
    assert (res[0] != 0x7FFF).sum() > 10000


def test_full_size_properties_default_pools(pkg, synth, gpu):
    """BASELINE frame size (640x480) and the upstream default pools (0x40000 blocks, 0x100000 + 0x20000 entries):
    properties that need no oracle -- structural invariants, allocation idempotence, integrate o de-integrate =
    identity, run-to-run determinism."""
    wl = synth.s_street()
    p = pkg.SceneParams(**wl.scene_kwargs)
    runs = []
    for rep in range(2):
        s = gpu.create_scene(p)
        rs = gpu.create_render_state(s, wl.W, wl.H)
        v = gpu.create_view(wl.W, wl.H)
        for i in range(3):
            rgba, mm, M = wl.frame(i)
            gpu.view_update(v, rgba, mm, timestamp=float(i))
            gpu.process_frame(s, v, rs, M, wl.intr)
        snap = util.snapshot(gpu, s, rs)
        snap["depth"] = gpu.get_image(s, rs, M, wl.intr, pkg.IMAGE_DEPTH)
        runs.append(snap)
        if rep == 0:
            util.check_invariants(snap, s.params)
            assert snap["stats"]["no_visible_entries"] > 5000
            # allocation idempotence: same frame again until collisions are resolved, then nothing new
            prev = None
            for it in range(8):
                gpu.allocate_scene_from_depth(s, v, rs, M, wl.intr)
                lf = gpu.stats(s, rs)["last_free_block_id"]
                if lf == prev:
                    break
                prev = lf
            assert it < 7
            ids = gpu.download_visible_ids(rs)
            assert (np.diff(ids) > 0).all()
        for o in (s, rs, v):
            o.close()
    util.assert_same_state(runs[0], runs[1], "determinism at full size")
    assert np.array_equal(runs[0]["depth"], runs[1]["depth"])
    hit = runs[0]["depth"] > 0
    true = wl.frame(2)[1].astype(np.float32) / 1000.0
    ok = hit & (true > 0)
    assert ok.mean() > 0.5 and np.median(np.abs(runs[0]["depth"] - true)[ok]) < 0.5 * wl.scene_kwargs["voxel_size"]
    # integrate o de-integrate = identity on a fresh map (weights never clamp at 1 observation)
    s = gpu.create_scene(p)
    rs = gpu.create_render_state(s, wl.W, wl.H)
    v = gpu.create_view(wl.W, wl.H)
    rgba, mm, M = wl.frame(0)
    gpu.view_update(v, rgba, mm)
    gpu.process_frame(s, v, rs, M, wl.intr)
    gpu.deprocess_frame(s, v, rs, M, wl.intr)
    vox = gpu.download_voxel_blocks(s, 0x40000 - 9000, 9000).view(np.uint64)
    assert (vox == 0x7FFF).all()


@pytest.mark.parametrize("budget", [40, 300, 1500])
def test_render_tile_budget_rule(pkg, synth, gpu, oracle, budget):
    """CreateExpectedDepths drops, in visible-list order, every block whose render tiles would reach
    MAX_RENDERING_BLOCKS (a dropped block does not count, so later smaller blocks still get in).  Real scenes never
    reach the 262144-tile budget; the test lowers it through the debug hook on both engines and compares the range
    image and the raycast that follows, through both call sequences (GetImage and the separate calls)."""
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl)
    imgs, ranges = {}, {}
    for name, api in (("gpu", gpu), ("oracle", oracle)):
        s = api.create_scene(p)
        rs, v = api.create_render_state(s, wl.W, wl.H), api.create_view(wl.W, wl.H)
        for i in range(4):
            rgba, mm, M = wl.frame(i)
            api.view_update(v, rgba, mm, timestamp=float(i))
            api.process_frame(s, v, rs, M, wl.intr)
        api.debug_set_render_tile_budget(budget)
        try:
            free = api.create_render_state(s, wl.W, wl.H)
            imgs[name] = api.get_image(s, free, M, wl.intr, pkg.IMAGE_DEPTH)
            # the tracking path: CreateExpectedDepths on the fusion render state's visible list
            api.create_expected_depths(s, rs, M, wl.intr)
            ranges[name] = api.download_range_image(rs)
            n_vis = api.stats(s, rs)["no_visible_entries"]
        finally:
            api.debug_set_render_tile_budget(65536 * 4)
        if name == "oracle":
            full = api.get_image(s, free, M, wl.intr, pkg.IMAGE_DEPTH)
    cw, ch = (wl.W + 7) // 8, (wl.H + 7) // 8
    assert np.array_equal(ranges["gpu"][:ch, :cw], ranges["oracle"][:ch, :cw])
    assert np.array_equal(imgs["gpu"] > 0, imgs["oracle"] > 0)
    assert np.abs(imgs["gpu"] - imgs["oracle"]).max() <= 1e-4
    assert n_vis > 300  # budgets 40 and 300 bite, 1500 does not
    if budget == 40:
        assert (imgs["gpu"] > 0).sum() < (full > 0).sum(), "a budget far below the visible count must lose surface"


@pytest.mark.parametrize("swapping", [False, True])
def test_full_size_parity_with_oracle(pkg, synth, gpu, oracle, swapping):
    """BASELINE configs[1]/[2] geometry at full size: 640x480 S-street frames, the upstream default pools
    (0x40000 voxel blocks, 0x100000 + 0x20000 hash entries), fusion + decay + sliding window, then the raycast --
    the whole map state byte-for-byte against the CPU oracle, not just the down-sized scenes of the other tests."""
    oracle.set_threads(16)
    wl = synth.s_street()
    p = pkg.SceneParams(use_swapping=int(swapping), **wl.scene_kwargs)
    n_frames, max_age = (14, 5) if swapping else (8, 5)
    objs = {}
    for name, api in (("gpu", gpu), ("oracle", oracle)):
        s = api.create_scene(p)
        objs[name] = (api, s, api.create_render_state(s, wl.W, wl.H), api.create_view(wl.W, wl.H))
    for i in range(n_frames):
        rgba, mm, M = wl.frame(i)
        for name, (api, s, rs, v) in objs.items():
            api.view_update(v, rgba, mm, timestamp=float(i))
            api.process_frame(s, v, rs, M, wl.intr)
            if api.stats(s, rs)["fusion_fifo_len"] > max_age:
                api.slide_window(s, rs, max_age)
            api.decay(s, rs, 1, 3, True)
        if i in (0, n_frames - 1):  # (a full snapshot moves 2 x 1 GiB of voxel blocks per engine)
            snaps = {name: util.snapshot(api, s, rs) for name, (api, s, rs, v) in objs.items()}
            util.assert_same_state(snaps["gpu"], snaps["oracle"], f"full size, frame {i}")
            if swapping:  # ITMGlobalCache bookkeeping and the blocks parked on the host
                sw = {name: api.download_swap_states(s) for name, (api, s, rs, v) in objs.items()}
                assert np.array_equal(sw["gpu"], sw["oracle"]), f"swap states, frame {i}"
                out = np.nonzero(snaps["gpu"]["hash"]["ptr"] == -1)[0]
                for t in out[:: max(1, len(out) // 40)]:
                    a, ba = gpu.download_stored_block(objs["gpu"][1], int(t))
                    b, bb = oracle.download_stored_block(objs["oracle"][1], int(t))
                    assert a == b and (not a or np.array_equal(ba.view(np.uint64), bb.view(np.uint64)))
    st = snaps["gpu"]["stats"]
    assert st["no_visible_entries"] > 5000
    if not swapping:
        assert st["slid_block_count"] + st["decayed_block_count"] > 0
    if swapping:  # (blocks that leave the view are parked on the host before the window or the decay reaches them)
        assert (snaps["gpu"]["hash"]["ptr"] == -1).sum() > 0, "something must have been swapped out"
    util.check_invariants(snaps["gpu"], objs["gpu"][1].params)
    del snaps
    imgs = {}
    for name, (api, s, rs, v) in objs.items():
        free = api.create_render_state(s, wl.W, wl.H)
        imgs[name] = (api.get_image(s, free, M, wl.intr, pkg.IMAGE_DEPTH),
                      api.get_image(s, free, M, wl.intr, pkg.IMAGE_COLOUR_FROM_VOLUME),
                      api.download_range_image(free)[:(wl.H + 7) // 8, :(wl.W + 7) // 8])
    assert np.array_equal(imgs["gpu"][2], imgs["oracle"][2]), "range image"
    assert np.array_equal(imgs["gpu"][0] > 0, imgs["oracle"][0] > 0) and (imgs["gpu"][0] > 0).mean() > 0.5
    assert np.abs(imgs["gpu"][0] - imgs["oracle"][0]).max() <= 1e-4  # metres (float; bit-identical in practice)
    assert np.abs(imgs["gpu"][1].astype(int) - imgs["oracle"][1].astype(int)).max() <= 1


def test_sharding_a_swapping_scene_is_refused(pkg, synth, gpu, oracle):
    """The sharded batch exchanges device blocks only; with host swapping every rank would also swap its own, partly
    stale copies out to its own host store (found by the fuzz test: a sharded re-fusion on a swapping scene left the
    replicas' maps different from the unsharded one).  Both engines refuse the combination instead."""
    wl = synth.s_tiny()
    for api in (gpu, oracle):
        s = api.create_scene(util.small_params(pkg, wl, use_swapping=1))
        with pytest.raises(pkg.DslamError):
            api.set_shard(s, 0, 2, 16)
        with pytest.raises(pkg.DslamError):
            api.set_shard_range(s, 0, 64)
        api.set_shard(s, 0, 1, 16)      # "not sharded" stays legal
        api.set_shard_range(s, 0, -1)
        s2 = api.create_scene(util.small_params(pkg, wl))
        api.set_shard(s2, 1, 2, 16)     # and a scene without swapping shards as before


@pytest.mark.parametrize("maintenance", [False, True])
def test_sharded_reintegration_exchanges_dirty_blocks(pkg, synth, gpu, oracle, maintenance):
    """The multi-GPU re-integration scheme rehearsed on one GPU: `world` map replicas play the ranks, each
    de-/re-integrates only its own slot chunks, the exchange is dslam_shard_dirty_plan -> dslam_shard_dirty_pack -> (what
    an all-gather delivers) -> dslam_shard_dirty_unpack.  Every replica must end byte-identical to the unsharded run --
    also on a map that was decayed and slid first (BASELINE configs[2]), where freed slots have gone back to the pool in
    arbitrary order and live blocks no longer form a top range of it -- and the lists must be the oracle's."""
    torch = pytest.importorskip("torch")
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl, num_local_blocks=0x800, num_buckets=0x1000, num_excess=0x400)
    world, chunk = 4, 16
    n_frames = 12 if maintenance else 6
    fix = (9, 10, 11) if maintenance else (1, 3, 4)
    new_M = {i: synth.world_to_camera(wl.pose(i) @ synth.pose_matrix(synth.look_rotation(0.004, 0.002), [0.004, -0.002, 0.003])) for i in fix}

    def build(api):
        if maintenance:
            return util.run_sequence(api, pkg, wl, p, n_frames, decay=(1, 2, True), slide=2)
        return util.run_sequence(api, pkg, wl, p, n_frames)

    def correct(api, s, rs, v):
        for i in fix:
            rgba, mm, M = wl.frame(i)
            api.view_update(v, rgba, mm, timestamp=float(i))
            api.deprocess_frame(s, v, rs, M, wl.intr)
            api.process_frame(s, v, rs, new_M[i], wl.intr, is_defusion=True)

    ref = build(gpu)
    before = util.snapshot(gpu, ref[0], ref[1])
    correct(gpu, *ref)
    want = util.snapshot(gpu, ref[0], ref[1])
    if maintenance:  # the case the slot-range exchange of round 1 got wrong
        used = p.num_local_blocks - 1 - before["stats"]["last_free_block_id"]
        assert int(before["hash"]["ptr"][before["hash"]["ptr"] >= 0].min()) < p.num_local_blocks - used

    ranks = [build(gpu) for _ in range(world)]
    counts = None
    for r, (s, rs, v) in enumerate(ranks):
        gpu.track_dirty(s, True)
        gpu.set_shard(s, r, world, chunk)
        correct(gpu, s, rs, v)
        c = gpu.shard_dirty_plan(s, world, chunk)
        assert counts is None or c == counts, "ranks disagree about the dirty lists"
        counts = c
    # the oracle, unsharded but tracking, names the same blocks
    so, rso, vo = build(oracle)
    oracle.track_dirty(so, True)
    correct(oracle, so, rso, vo)
    assert oracle.shard_dirty_plan(so, world, chunk) == counts and min(counts) > 20
    cap = max(counts)
    recv = torch.zeros((world, cap, 4096), dtype=torch.uint8, device="cuda")
    for r, (s, rs, v) in enumerate(ranks):  # every rank's send buffer lands in slice r of every rank's recv buffer
        gpu.shard_dirty_pack(s, r, recv[r].data_ptr(), cap)
    gpu.synchronize()
    for r, (s, rs, v) in enumerate(ranks):
        if r == 0:  # before the exchange a rank holds only its own shard's updates
            assert not np.array_equal(gpu.download_voxel_blocks(s).view(np.uint64), want["voxels"].view(np.uint64))
        gpu.shard_dirty_unpack(s, r, recv.data_ptr(), cap)
        gpu.set_shard(s, 0, 1, chunk)
        gpu.track_dirty(s, False)
        util.assert_same_state(util.snapshot(gpu, s, rs), want, f"rank {r} after the exchange")
    changed = int((want["voxels"].view(np.uint64) != before["voxels"].view(np.uint64)).any(axis=1).sum())
    assert sum(counts) >= changed > 200


@pytest.mark.parametrize("size", [(912, 228), (1226, 370), (70, 45)])
def test_kitti_frame_sizes_and_partial_tiles(pkg, synth, gpu, oracle, size):
    """The frame sizes of BASELINE configs 2, 3 and 5 (KITTI crop 912x228, full 1226x370; DenseSLAMGUI.cpp:504,
    scripts/eval_raycast_depth.py:90-92) and a small odd one.  1226, 370, 228, 70 and 45 are not multiples of 8: the
    last 8x8 ray tile of a row / column is partial and the range image keeps upstream's full-image stride.  Fusion
    with decay and a sliding window, then every raycast product, against the oracle."""
    W, H = size
    oracle.set_threads(16)
    big = W > 100
    wl = synth.s_street(W, H) if big else synth.s_room(W, H, scale=4.0)
    kw = dict(num_local_blocks=0x10000) if big else dict(num_local_blocks=0x800, num_buckets=0x1000, num_excess=0x400)
    p = pkg.SceneParams(**kw, **wl.scene_kwargs)
    objs = {}
    for name, api in (("gpu", gpu), ("oracle", oracle)):
        s = api.create_scene(p)
        objs[name] = (api, s, api.create_render_state(s, W, H), api.create_view(W, H), api.create_render_state(s, W, H))
    n_frames = 5
    for i in range(n_frames):
        rgba, mm, M = wl.frame(i)
        for name, (api, s, rs, v, free) in objs.items():
            api.view_update(v, rgba, mm, timestamp=float(i))
            api.process_frame(s, v, rs, M, wl.intr)
            if api.stats(s, rs)["fusion_fifo_len"] > 3:
                api.slide_window(s, rs, 3)
            api.decay(s, rs, 1, 2, True)
    snaps = {name: util.snapshot(api, s, rs) for name, (api, s, rs, v, free) in objs.items()}
    util.assert_same_state(snaps["gpu"], snaps["oracle"], f"{W}x{H}")
    assert snaps["gpu"]["stats"]["no_visible_entries"] > (3000 if big else 50)
    M = wl.frame(n_frames - 1)[2]
    out = {}
    for name, (api, s, rs, v, free) in objs.items():
        out[name] = dict(depth=api.get_image(s, free, M, wl.intr, pkg.IMAGE_DEPTH),
                         colour=api.get_image(s, free, M, wl.intr, pkg.IMAGE_COLOUR_FROM_VOLUME),
                         shaded=api.get_image(s, free, M, wl.intr, pkg.IMAGE_SHADED),
                         rng=api.download_range_image(free), icp=api.create_icp_maps(s, rs, M, wl.intr),
                         mm=api.get_depth_image_int16(s, free, M, wl.intr, 256))
    g, o = out["gpu"], out["oracle"]
    assert np.array_equal(g["rng"][:(H + 7) // 8, :(W + 7) // 8], o["rng"][:(H + 7) // 8, :(W + 7) // 8])
    assert np.array_equal(g["depth"] > 0, o["depth"] > 0) and (g["depth"] > 0).mean() > 0.3
    assert np.abs(g["depth"] - o["depth"]).max() <= 1e-4
    # the partial tiles at the right and bottom edges are rendered like any other
    assert W % 8 == 0 or (g["depth"][:, (W // 8) * 8:] > 0).any()
    assert H % 8 == 0 or (g["depth"][(H // 8) * 8:, :] > 0).any()
    assert np.abs(g["mm"].astype(int) - o["mm"].astype(int)).max() <= 1
    for k in ("colour", "shaded"):
        assert np.abs(g[k].astype(int) - o[k].astype(int)).max() <= 1, k
    assert np.array_equal(g["icp"][0][..., 3], o["icp"][0][..., 3])
    assert np.abs(g["icp"][0] - o["icp"][0]).max() <= 1e-4 and np.abs(g["icp"][1] - o["icp"][1]).max() <= 1e-4


def test_get_image_memo_same_view_is_shaded_only_and_invalidated_by_any_change(pkg, synth, gpu, oracle):
    """The reference's GUI asks for a depth and a colour image of the same free pose every tick (DenseSlam.h:146-164).
    The engine keeps the march of the last GetImage per render state and only shades when scene version, pose and
    intrinsics are unchanged.  Every image type from the memo must equal a fresh render, and anything that can change
    the result -- fusing a frame, decay, another pose, another user of the render state -- must drop the memo."""
    wl = synth.s_tiny(96, 72)
    p = util.small_params(pkg, wl)
    objs = {}
    for name, api in (("gpu", gpu), ("oracle", oracle)):
        s, rs, v = util.run_sequence(api, pkg, wl, p, 3)
        objs[name] = (api, s, rs, v, api.create_render_state(s, wl.W, wl.H), api.create_render_state(s, wl.W, wl.H))
    M0, M1 = wl.frame(2)[2], wl.frame(1)[2]
    types = (pkg.IMAGE_DEPTH, pkg.IMAGE_COLOUR_FROM_VOLUME, pkg.IMAGE_SHADED, pkg.IMAGE_COLOUR_FROM_NORMAL)

    def same(a, b, what):
        if a.dtype == np.float32:
            assert np.abs(a - b).max() <= 1e-4, what
        else:
            assert np.abs(a.astype(int) - b.astype(int)).max() <= 1, what

    g, s, rs, v, free, fresh = objs["gpu"]
    o, so, rso, vo, freeo, _ = objs["oracle"]
    # (1) four types of one view through one render state (memo) == each through a just-created render state == oracle
    for t in types:
        memo = g.get_image(s, free, M0, wl.intr, t)
        cold = g.get_image(s, g.create_render_state(s, wl.W, wl.H), M0, wl.intr, t)
        assert np.array_equal(memo, cold), f"type {t}: memo differs from a fresh render"
        same(memo, o.get_image(so, freeo, M0, wl.intr, t), f"type {t} vs oracle")
    assert np.array_equal(g.get_depth_image_int16(s, free, M0, wl.intr, 1000), o.get_depth_image_int16(so, freeo, M0, wl.intr, 1000))
    # (2) another pose, then back: both rendered for what they are
    same(g.get_image(s, free, M1, wl.intr, pkg.IMAGE_DEPTH), o.get_image(so, freeo, M1, wl.intr, pkg.IMAGE_DEPTH), "other pose")
    same(g.get_image(s, free, M0, wl.intr, pkg.IMAGE_DEPTH), o.get_image(so, freeo, M0, wl.intr, pkg.IMAGE_DEPTH), "back")
    # (3) the map changes between two requests for the same view
    before = g.get_image(s, free, M0, wl.intr, pkg.IMAGE_DEPTH)
    rgba, mm, M3 = wl.frame(3)
    for api, sc, r, vw in ((g, s, rs, v), (o, so, rso, vo)):
        api.view_update(vw, rgba, mm, timestamp=3.0)
        api.process_frame(sc, vw, r, M3, wl.intr)
    after = g.get_image(s, free, M0, wl.intr, pkg.IMAGE_DEPTH)
    same(after, o.get_image(so, freeo, M0, wl.intr, pkg.IMAGE_DEPTH), "after fusing a frame")
    assert not np.array_equal(before, after), "a fused frame must show in the next image of the same view"
    for api, sc, r in ((g, s, rs), (o, so, rso)):
        api.decay(sc, r, 2, 0, True)
    same(g.get_image(s, free, M0, wl.intr, pkg.IMAGE_DEPTH), o.get_image(so, freeo, M0, wl.intr, pkg.IMAGE_DEPTH), "after decay")
    # (4) another user of the render state in between (tracking raycast from a different pose)
    g.get_image(s, free, M0, wl.intr, pkg.IMAGE_DEPTH)
    g.create_icp_maps(s, free, M1, wl.intr)
    o.create_icp_maps(so, freeo, M1, wl.intr)
    same(g.get_image(s, free, M0, wl.intr, pkg.IMAGE_SHADED), o.get_image(so, freeo, M0, wl.intr, pkg.IMAGE_SHADED), "after ICP maps")
    # (5) different intrinsics
    intr2 = np.array(wl.intr, np.float32) * np.float32(0.9)
    same(g.get_image(s, free, M0, intr2, pkg.IMAGE_DEPTH), o.get_image(so, freeo, M0, intr2, pkg.IMAGE_DEPTH), "other intrinsics")


def test_get_image_into_page_locked_caller_image(pkg, synth, gpu):
    """A page-locked output image (dslam_host_alloc, what the ITMLib mirror's images are) is written by the render
    kernel itself instead of by a copy queued behind it: same pixels as the ordinary path, for float and RGBA images,
    from a fresh march and from the GetImage memo, and for a pinned buffer that is larger than the image."""
    wl = synth.s_tiny(96, 72)
    p = util.small_params(pkg, wl)
    s, rs, v = util.run_sequence(gpu, pkg, wl, p, 3)
    free = gpu.create_render_state(s, wl.W, wl.H)
    M0, M1 = wl.frame(2)[2], wl.frame(1)[2]
    pin_f = gpu.host_alloc((wl.H, wl.W), np.float32)
    pin_c = gpu.host_alloc((wl.H + 3, wl.W, 4), np.uint8)
    try:
        want_f = gpu.get_image(s, gpu.create_render_state(s, wl.W, wl.H), M0, wl.intr, pkg.IMAGE_DEPTH)
        want_c = gpu.get_image(s, gpu.create_render_state(s, wl.W, wl.H), M0, wl.intr, pkg.IMAGE_COLOUR_FROM_VOLUME)
        pin_f[...] = -7.0
        got = gpu.get_image(s, free, M0, wl.intr, pkg.IMAGE_DEPTH, out=pin_f)      # fresh march, direct store
        assert got is pin_f and np.array_equal(pin_f, want_f) and (want_f > 0).mean() > 0.3
        pin_c[...] = 9
        gpu.get_image(s, free, M0, wl.intr, pkg.IMAGE_COLOUR_FROM_VOLUME, out=pin_c[:wl.H])  # from the memo, direct store
        assert np.array_equal(pin_c[:wl.H], want_c) and (pin_c[wl.H:] == 9).all()
        # the device copy of the image is not needed afterwards: another view, ordinary output, still right
        other = gpu.get_image(s, free, M1, wl.intr, pkg.IMAGE_DEPTH)
        assert np.array_equal(other, gpu.get_image(s, gpu.create_render_state(s, wl.W, wl.H), M1, wl.intr, pkg.IMAGE_DEPTH))
    finally:
        gpu.host_free(pin_f)
        gpu.host_free(pin_c)


def test_range_image_with_blocks_next_to_the_camera(pkg, synth, gpu, oracle):
    """Free-camera views from a few centimetres off the fused surface: the nearest blocks project onto boxes that cover whole
    16x16-cell tiles of the range image -- the boxes k_fill_range_tiles hands to the whole workgroup (thread = cell) instead of
    splatting them from one lane (round 3) -- next to thousands of small ones.  Range image (the corner the raycaster reads)
    and depth image against the oracle, 640x480."""
    wl = synth.s_street(640, 480)
    p = pkg.SceneParams(num_local_blocks=0x8000, **wl.scene_kwargs)
    out = {}
    for name, api in (("gpu", gpu), ("oracle", oracle)):
        s = api.create_scene(p)
        rs, v = api.create_render_state(s, wl.W, wl.H), api.create_view(wl.W, wl.H)
        for i in range(3):
            rgba, mm, M = wl.frame(i)
            api.view_update(v, rgba, mm, timestamp=float(i))
            api.process_frame(s, v, rs, M, wl.intr)
        free = api.create_render_state(s, wl.W, wl.H)
        res = []
        for drop, pitch in ((1.3, -0.5), (1.45, -0.3), (1.48, 0.0)):   # metres towards the ground, looking down by -`pitch` rad
            T = wl.pose(2) @ synth.pose_matrix(synth.look_rotation(0.0, pitch), [0.0, drop, 0.0])
            Mf = synth.world_to_camera(T)
            img = api.get_image(s, free, Mf, wl.intr, pkg.IMAGE_DEPTH)
            res.append((img, api.download_range_image(free)[:(wl.H + 7) // 8, :(wl.W + 7) // 8].copy(), api.stats(s, free)["no_visible_entries"]))
        out[name] = res
    big_tiles = 0
    for k, ((gi, gr, gn), (oi, orr, on)) in enumerate(zip(out["gpu"], out["oracle"])):
        assert gn == on and gn > 1000, f"view {k}: {gn} / {on} visible blocks"
        assert np.array_equal(gr, orr), f"view {k}: range image"
        assert np.array_equal(gi > 0, oi > 0) and np.abs(gi - oi).max() <= 1e-4, f"view {k}: depth image"
        valid = orr[..., 1] > orr[..., 0]
        assert orr[..., 0][valid].min() < 1.0, f"view {k}: nothing closer than a metre"
        big_tiles += int((orr[..., 0][valid] < 1.5).sum())
    # a 0.4 m block at 0.7 m spans ~300 pixels = 38 range-image cells: more than two 16-cell tiles in each direction
    assert big_tiles > 500, "the views must contain many cells whose nearest block is closer than 1.5 m"
