"""Parity of the on-device meshing export with the CPU oracle, through the C ABI (SaveCurrSceneToMesh ->
ITMMeshingEngine::MeshScene, reference DenseSlam.cpp:638-643).  The triangle list is compared bit
for bit, order included: the engine reproduces the CPU engine's order with scans instead of upstream CUDA's atomics."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


def _both(gpu, oracle):
    return (("gpu", gpu), ("oracle", oracle))


def _same_mesh(a, b, what):
    assert a[0].shape == b[0].shape, f"{what}: {a[0].shape[0]} vs {b[0].shape[0]} triangles"
    assert np.array_equal(a[0], b[0]), f"{what}: positions"
    if a[1] is not None or b[1] is not None:
        assert np.array_equal(a[1], b[1]), f"{what}: colours"


@pytest.mark.parametrize("colour", [False, True])
def test_mesh_bit_exact_tiny(pkg, synth, gpu, oracle, colour):
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl)
    out = {}
    for name, api in _both(gpu, oracle):
        s, rs, v = util.run_sequence(api, pkg, wl, p, 4)
        out[name] = api.mesh_scene(s, colour=colour)
        if name == "gpu":
            again = api.mesh_scene(s, colour=colour)  # deterministic: no atomics in the ordering
            _same_mesh(out[name], again, "second run")
    assert len(out["gpu"][0]) > 10000
    _same_mesh(out["gpu"], out["oracle"], "tiny")


def test_mesh_after_decay_and_slide(pkg, synth, gpu, oracle):
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl)
    out = {}
    for name, api in _both(gpu, oracle):
        s, rs, v = util.run_sequence(api, pkg, wl, p, 8, decay=(2, 2, True), slide=3)
        out[name] = api.mesh_scene(s, colour=True)
    assert len(out["oracle"][0]) > 1000
    _same_mesh(out["gpu"], out["oracle"], "after decay + slide")


def test_mesh_with_swapped_out_blocks(pkg, synth, gpu, oracle):
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl, use_swapping=1)
    out, swapped = {}, {}
    for name, api in _both(gpu, oracle):
        s, rs, v = util.run_sequence(api, pkg, wl, p, 3)
        far = wl.frame(0)[2].copy()
        far[:3, 3] += np.array([30.0, 0.0, 0.0], np.float32)
        rgba, mm, _ = wl.frame(0)
        for i in range(3):
            api.view_update(v, rgba, np.zeros_like(mm), timestamp=10.0 + i)
            api.process_frame(s, v, rs, far, wl.intr)
        swapped[name] = int((api.download_hash_table(s)["ptr"] == -1).sum())
        out[name] = api.mesh_scene(s, colour=True)
    assert swapped["gpu"] == swapped["oracle"] > 0
    _same_mesh(out["gpu"], out["oracle"], "with swapped-out blocks")


def test_mesh_saturation_and_empty(pkg, synth, gpu, oracle):
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl)
    empty = gpu.mesh_scene(gpu.create_scene(p), colour=True)
    assert empty[0].shape == (0, 3, 3) and empty[1].shape == (0, 3, 3)
    res = {}
    for name, api in _both(gpu, oracle):
        s, rs, v = util.run_sequence(api, pkg, wl, p, 2)
        res[name] = [api.mesh_scene(s, max_triangles=m) for m in (0, 100, 1, 2, 5000)]
    for g, o, m in zip(res["gpu"], res["oracle"], (0, 100, 1, 2, 5000)):
        _same_mesh(g, o, f"max_triangles={m}")
    assert len(res["gpu"][1][0]) == 99 and len(res["gpu"][2][0]) == 0 and len(res["gpu"][3][0]) == 1
    # a mesh made without colours cannot be downloaded with them
    with pytest.raises(Exception):
        import ctypes as C
        buf = np.empty((len(res["gpu"][4][0]) + 1, 3, 3), np.float32)
        gpu._call("mesh_download", gpu._engine, buf.ctypes.data_as(C.POINTER(C.c_float)),
                  buf.ctypes.data_as(C.POINTER(C.c_float)), C.c_int(len(buf)))


def test_mesh_full_size_pools_street(pkg, synth, gpu, oracle):
    """S-street at the metric's frame size with the reference's default pools (0x40000 blocks, 0x100000 buckets):
    ~8 k blocks, a live list spread over the whole 1.18 M-entry table, excess-list neighbours."""
    wl = synth.s_street(640, 480)
    p = pkg.SceneParams(**wl.scene_kwargs)
    out = {}
    for name, api in _both(gpu, oracle):
        s, rs, v = util.run_sequence(api, pkg, wl, p, 3)
        out[name] = api.mesh_scene(s, colour=True)
    assert len(out["oracle"][0]) > 100000
    _same_mesh(out["gpu"], out["oracle"], "S-street 640x480")
