"""Parity of the on-device view pre-processing with the CPU oracle, through the C ABI:
BGR->RGBA (reference InfiniTamDriver.cpp:84-103), the bilateral depth filter of ITMViewBuilder::UpdateView
(call site InfiniTamDriver.cpp:280-288) and DenseSlam::depthPostProcessing (DenseSlam.cpp:434-552).
Integer / byte results are bit-exact; the filtered float depth is bit-exact too because host and device share one
fixed exp sequence (tolerance 0)."""
import ctypes

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(7, 5), (64, 48), (641, 3)])
def test_bgr_to_rgba_bit_exact(gpu, oracle, shape):
    W, H = shape
    rng = np.random.default_rng(W * 1000 + H)
    bgr = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    mm = rng.integers(0, 4000, (H, W)).astype(np.int16)
    out = {}
    for name, api in (("gpu", gpu), ("oracle", oracle)):
        v = api.create_view(W, H)
        api.view_update_bgr(v, bgr, mm)
        out[name] = (api.download_view_rgba(v), api.download_view_depth(v))
    assert np.array_equal(out["gpu"][0], out["oracle"][0])
    assert np.array_equal(out["gpu"][1], out["oracle"][1])


def test_bgr_device_resident(gpu, oracle):
    torch = pytest.importorskip("torch")
    W, H = 64, 48
    rng = np.random.default_rng(5)
    bgr = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    mm = rng.integers(0, 4000, (H, W)).astype(np.int16)
    t_bgr, t_mm = torch.from_numpy(bgr).cuda(), torch.from_numpy(mm).cuda()
    torch.cuda.synchronize()
    v = gpu.create_view(W, H)
    gpu.view_update_bgr_device(v, t_bgr.data_ptr(), t_mm.data_ptr())
    vo = oracle.create_view(W, H)
    oracle.view_update_bgr(vo, bgr, mm)
    assert np.array_equal(gpu.download_view_rgba(v), oracle.download_view_rgba(vo))


def test_bilateral_filter_bit_exact_and_fused_state(pkg, synth, gpu, oracle):
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl)
    objs = {}
    for name, api in (("gpu", gpu), ("oracle", oracle)):
        s = api.create_scene(p)
        objs[name] = (api, s, api.create_render_state(s, wl.W, wl.H), api.create_view(wl.W, wl.H))
    rng = np.random.default_rng(11)
    for i in range(4):
        rgba, mm, M = wl.frame(i)
        noisy = mm.astype(np.int32) + rng.integers(-4, 5, mm.shape)
        noisy[mm <= 0] = 0
        noisy[5:9, 7:12] = 0  # a hole
        noisy = noisy.astype(np.int16)
        snaps, depths = {}, {}
        for name, (api, s, rs, v) in objs.items():
            api.view_update(v, rgba, noisy, timestamp=float(i), bilateral=True)
            depths[name] = api.download_view_depth(v)
            api.process_frame(s, v, rs, M, wl.intr)
            snaps[name] = util.snapshot(api, s, rs)
        assert np.array_equal(depths["gpu"].view(np.uint32), depths["oracle"].view(np.uint32)), f"frame {i}: filtered depth"
        util.assert_same_state(snaps["gpu"], snaps["oracle"], f"frame {i}")
    assert snaps["gpu"]["stats"]["no_visible_entries"] > 50
    # switching the filter off again gives the plain conversion (scratch buffers swapped back correctly)
    rgba, mm, M = wl.frame(4)
    for name, (api, s, rs, v) in objs.items():
        api.view_update(v, rgba, mm, timestamp=4.0)
        depths[name] = api.download_view_depth(v)
    assert np.array_equal(depths["gpu"], depths["oracle"])
    assert np.array_equal(depths["gpu"][mm > 0], (mm[mm > 0].astype(np.float32) * np.float32(0.001)))


def test_bilateral_filter_full_size(gpu, oracle, synth):
    wl = synth.s_room()
    rgba, mm, M = wl.frame(3)
    out = {}
    for name, api in (("gpu", gpu), ("oracle", oracle)):
        v = api.create_view(wl.W, wl.H)
        api.view_update(v, rgba, mm, bilateral=True)
        out[name] = api.download_view_depth(v)
    assert np.array_equal(out["gpu"].view(np.uint32), out["oracle"].view(np.uint32))


def _pose_delta(synth, ang, t):
    return synth.pose_matrix(synth.look_rotation(ang, 0.5 * ang), t).astype(np.float32)


@pytest.mark.parametrize("case", ["identity", "small_motion", "large_motion", "behind_camera"])
def test_depth_post_processing_bit_exact(gpu, oracle, synth, case):
    wl = synth.s_street(912, 228)
    _, curr, _ = wl.frame(1)
    _, prev, _ = wl.frame(0)
    rng = np.random.default_rng(7)
    curr = curr.copy()
    curr[rng.random(curr.shape) < 0.02] = 0
    curr[3, 4] = -5
    prev = prev.copy()
    prev[rng.random(prev.shape) < 0.02] = 0
    prev[7, 9] = -1  # read back as uint16 65535 -> 65.535 m
    Tpc = {"identity": np.eye(4, dtype=np.float32),
           "small_motion": _pose_delta(synth, 0.01, [0.02, -0.01, 0.3]),
           "large_motion": _pose_delta(synth, 0.1, [0.3, 0.1, -0.5]),
           "behind_camera": _pose_delta(synth, 3.0, [0.0, 0.0, -60.0])}[case]
    # the reference feeds (fx, fy, cx, cy) of the colour camera; rows pair with cx -- use the workload's intrinsics
    g, gc = gpu.depth_post_processing(curr, prev, Tpc, wl.intr, 0.05, 0.3)
    o, oc = oracle.depth_post_processing(curr, prev, Tpc, wl.intr, 0.05, 0.3)
    assert np.array_equal(g, o) and gc == oc
    if case == "small_motion":
        assert gc > 1000


def test_depth_post_processing_device_resident(gpu, oracle, synth):
    torch = pytest.importorskip("torch")
    wl = synth.s_room()
    _, curr, _ = wl.frame(2)
    _, prev, _ = wl.frame(0)
    Tpc = _pose_delta(synth, 0.02, [0.05, 0.0, 0.1])
    intr = (wl.intr[1], wl.intr[0], wl.intr[3], wl.intr[2])  # swapped so that projections land inside the image
    o, oc = oracle.depth_post_processing(curr, prev, Tpc, intr, 0.02, 0.0)
    tc, tp = torch.from_numpy(curr.copy()).cuda(), torch.from_numpy(prev.copy()).cuda()
    torch.cuda.synchronize()
    gc = gpu.depth_post_processing_device(tc.data_ptr(), tp.data_ptr(), wl.W, wl.H, Tpc, intr, 0.02, 0.0)
    gpu.synchronize()
    assert gc == oc and np.array_equal(tc.cpu().numpy(), o)
    assert (o != curr).sum() > 0, "the case should blank something"


@pytest.mark.parametrize("fmt,max_m", [(1, 40.0), (1, 20.0), (2, 10.0), (2, 4.0), (0, 0.0)])
def test_dataset_depth_formats_bit_exact(gpu, oracle, synth, fmt, max_m):
    """PrecomputedDepthProvider::ReadPrecomputed's loop (PrecomputedDepthProvider.cpp:30-64) on the device: raw 16-bit
    dataset depth -> millimetres, incl. the values the reference's int16 casts wrap."""
    wl = synth.s_street(320, 96)
    rgba, _, _ = wl.frame(0)
    rng = np.random.default_rng(fmt * 10 + int(max_m))
    raw = rng.integers(-300, 32768, (wl.H, wl.W)).astype(np.int16)  # every positive int16 and a few negatives
    raw[0, :8] = [0, 1, 255, 256, 5119, 5120, 10240, 10241]
    out = {}
    for name, api in (("gpu", gpu), ("oracle", oracle)):
        v = api.create_view(wl.W, wl.H)
        api.view_update_dataset(v, rgba, raw, fmt, max_m)
        out[name] = (api.download_view_raw_depth(v), api.download_view_depth(v), api.download_view_rgba(v))
        api.view_update_dataset(v, np.ascontiguousarray(rgba[..., 2::-1]), raw, fmt, max_m)  # BGR input, same depth
        assert np.array_equal(api.download_view_raw_depth(v), out[name][0])
    for a, b in zip(out["gpu"], out["oracle"]):
        assert np.array_equal(a, b)
    if fmt == 0:
        assert np.array_equal(out["gpu"][0], raw)
    if fmt == 2:
        assert out["gpu"][0].max() <= max_m * 1000


def test_depth_image_int16_output(pkg, synth, gpu, oracle):
    """FloatDepthmapToShort (x1000) and FloatDepthmapToInt16 (x256, the raycast-depth PNGs): InfiniTamDriver.cpp:167-200."""
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl)
    res = {}
    for name, api in (("gpu", gpu), ("oracle", oracle)):
        s = api.create_scene(p)
        rs, v = api.create_render_state(s, wl.W, wl.H), api.create_view(wl.W, wl.H)
        for i in range(3):
            rgba, mm, M = wl.frame(i)
            api.view_update(v, rgba, mm, timestamp=float(i))
            api.process_frame(s, v, rs, M, wl.intr)
        free = api.create_render_state(s, wl.W, wl.H)
        d = api.get_image(s, free, M, wl.intr, pkg.IMAGE_DEPTH)
        res[name] = (d, api.get_depth_image_int16(s, free, M, wl.intr, 1000), api.get_depth_image_int16(s, free, M, wl.intr, 256))
        for k, scale in ((1, 1000), (2, 256)):  # the int16 image is exactly the cast of the engine's own float image
            assert np.array_equal(res[name][k], (d * np.float32(scale)).astype(np.int32).astype(np.int16))
    assert np.abs(res["gpu"][0] - res["oracle"][0]).max() <= 1e-4
    assert np.abs(res["gpu"][1].astype(int) - res["oracle"][1].astype(int)).max() <= 1  # 1 mm
    assert (res["gpu"][2] > 0).sum() > 500


def test_page_locked_caller_images_take_the_direct_path(gpu, oracle):
    """dslam_host_alloc: the caller's image buffers page-locked (what upstream's MemoryBlock gives every image that
    has a device side).  In synchronous mode the upload DMAs straight out of them; in async mode, and from ordinary
    host memory, it is staged -- all three must leave the same view on the device, and the buffer may be rewritten
    right after the call returns."""
    W, H = 96, 64
    rng = np.random.default_rng(21)
    rgba_p = gpu.host_alloc((H, W, 4), np.uint8)
    mm_p = gpu.host_alloc((H, W), np.int16)
    assert not rgba_p.any() and not mm_p.any()  # zero-filled
    v, vo = gpu.create_view(W, H), oracle.create_view(W, H)
    try:
        for mode in (False, True, False):
            gpu.set_async(mode)
            for i in range(3):
                rgba = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
                mm = rng.integers(0, 4000, (H, W)).astype(np.int16)
                rgba_p[...] = rgba
                mm_p[...] = mm
                gpu.view_update(v, rgba_p, mm_p, timestamp=float(i))
                if not mode:  # the call has returned, so the buffers are the caller's again
                    rgba_p[...] = 0
                    mm_p[...] = -1
                oracle.view_update(vo, rgba, mm, timestamp=float(i))
                assert np.array_equal(gpu.download_view_rgba(v), oracle.download_view_rgba(vo))
                assert np.array_equal(gpu.download_view_depth(v), oracle.download_view_depth(vo))
        # a slice of a page-locked buffer is still inside the registered range; one past its end is not
        big = gpu.host_alloc((H * W * 4 + 64,), np.uint8)
        big[64:] = rgba.reshape(-1)
        gpu.view_update(v, big[64:], mm, timestamp=9.0)  # colour pinned, depth pageable -> staged
        assert np.array_equal(gpu.download_view_rgba(v), rgba)
        gpu.host_free(big)
    finally:
        gpu.set_async(False)
        gpu.host_free(rgba_p)
        gpu.host_free(mm_p)
    with pytest.raises(Exception):
        gpu._call("host_free", ctypes.c_void_p(12345))  # not one of ours
