"""Formats and metrics around the hot path: checked against the reference's own numbers/formulas."""
import math

import numpy as np
import pytest

import util


def test_memory_log_matches_reference_first_frame(pkg):
    """memory.txt:2 of the reference reads 0.324492 after the first fused keyframe = 8,307 blocks."""
    from dslam_amd.harness import evalio
    stats = {"num_allocated_blocks": 0x40000, "last_free_block_id": 0x40000 - 8307}
    line = evalio.memory_log_line(2, evalio.used_memory_bytes(stats))
    assert line == "2 0.324492"
    full = {"num_allocated_blocks": 0x40000, "last_free_block_id": 0x40000 - 261507}  # memory.txt:320
    assert evalio.memory_log_line(320, evalio.used_memory_bytes(full)) == "320 10.2151"


def test_depth_png_roundtrip_and_crop(pkg):
    from dslam_amd.harness import evalio
    d = np.array([[0.0, 1.0, 12.3456, 49.99]], np.float32)
    png = evalio.depth_to_png16(d)
    assert png.dtype == np.int16 and png.tolist() == [[0, 256, 3160, 12797]]
    assert np.allclose(evalio.png16_to_depth(png), d, atol=1 / 256)
    img = np.arange(370 * 1226).reshape(370, 1226)
    c = evalio.crop_bottom_centre(img)
    assert c.shape == (228, 912) and c[0, 0] == img[370 - 228, 157]


def test_depth_metrics_follow_the_reference_formulas(pkg):
    from dslam_amd.harness import evalio
    rng = np.random.RandomState(5)
    gt = rng.uniform(0.0, 60.0, size=(228, 912))
    pred = gt * rng.uniform(0.9, 1.1, size=gt.shape)
    pred[rng.rand(*gt.shape) < 0.1] = 0.0
    m = evalio.depth_metrics(pred, gt)
    # independent evaluation, line by line as scripts/eval_raycast_depth.py:100-136
    mask = (pred > 0.01) * (gt > 0.01) * (pred < 50) * (gt < 50)
    o, t = 1e3 * pred[mask], 1e3 * gt[mask]
    ad = np.abs(o - t)
    assert m["mask_number"] == np.count_nonzero(mask)
    assert math.isclose(m["mae"], ad.mean()) and math.isclose(m["rmse"], math.sqrt(float((ad * ad).mean())))
    assert math.isclose(m["absrel"], float((ad / t).mean())) and math.isclose(m["squared_rel"], float(((ad / t) ** 2).mean()))
    assert math.isclose(m["lg10"], float(np.abs(np.log10(o) - np.log10(t)).mean()))
    mr = np.maximum(o / t, t / o)
    assert math.isclose(m["delta1_125"], float((mr < 1.25).mean())) and math.isclose(m["delta3_101"], float((mr < 1.01 ** 3).mean()))


@pytest.mark.gpu
def test_memory_curves_keep_the_reference_ordering(pkg, synth, gpu):
    """The reference's four memory logs order as origin > decay > sliding window > sliding window + decay
    (memory.txt, memory_decay.txt, memory_slide_window.txt, memory_decay_slide_window.txt)."""
    from dslam_amd.harness import evalio
    wl = synth.s_street(320, 240)
    n_frames = 40
    used = {}
    for name, decay, slide in (("origin", None, None), ("decay", (2, 6, True), None), ("slide", None, 12),
                               ("both", (2, 6, True), 12)):
        p = pkg.SceneParams(num_local_blocks=0x10000, **wl.scene_kwargs)
        s, rs, v = util.run_sequence(gpu, pkg, wl, p, n_frames, decay=decay, slide=slide)
        st = gpu.stats(s, rs)
        st["num_allocated_blocks"] = p.num_local_blocks
        used[name] = evalio.used_memory_bytes(st)
        for o in (s, rs, v):
            o.close()
    assert used["origin"] > used["decay"] > used["both"]
    assert used["origin"] > used["slide"] > used["both"]
    # raycast accuracy on the fused map, measured like the reference's evaluation script
    p = pkg.SceneParams(num_local_blocks=0x10000, **wl.scene_kwargs)
    s, rs, v = util.run_sequence(gpu, pkg, wl, p, 10)
    rgba, mm, M = wl.frame(9)
    d = gpu.get_image(s, rs, M, wl.intr, pkg.IMAGE_DEPTH)
    m = evalio.depth_metrics(d, mm.astype(np.float64) / 1000.0)
    assert m["mask_number"] > 20000 and m["delta1_125"] > 0.95 and m["mae"] < 60.0  # mm, 5 cm voxels
