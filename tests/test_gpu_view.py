"""Parity of the on-device view pre-processing with the CPU oracle, through the C ABI:
BGR->RGBA (reference InfiniTamDriver.cpp:84-103), the bilateral depth filter of ITMViewBuilder::UpdateView
(call site InfiniTamDriver.cpp:280-288) and DenseSlam::depthPostProcessing (DenseSlam.cpp:434-552).
Integer / byte results are bit-exact; the filtered float depth is bit-exact too because host and device share one
fixed exp sequence (tolerance 0)."""
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
