"""CPU checks of the oracle's view pre-processing restatements: the deterministic exp, the
bilateral depth filter (upstream InfiniTAM v2 filterDepth; parity unpinned), CvToItm's BGR->RGBA, and
DenseSlam::depthPostProcessing (reference DenseSlam.cpp:434-552)."""
import ctypes as C

import numpy as np



def test_det_exp_within_one_ulp_of_correct_rounding(oracle):
    f = oracle.lib.oracle_det_exp
    f.restype, f.argtypes = C.c_float, [C.c_float]
    rng = np.random.default_rng(1)
    xs = np.concatenate([-rng.random(20000, dtype=np.float32) * 86, -rng.random(20000, dtype=np.float32) * 2,
                         np.float32([0, -1e-8, -86, -85.99, -0.5, -40])]).astype(np.float32)
    got = np.array([f(float(x)) for x in xs], dtype=np.float32)
    want = np.exp(xs.astype(np.float64)).astype(np.float32)
    ulp = np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
    assert ulp.max() <= 1
    assert f(0.0) == 1.0 and f(-86.5) == 0.0 and f(-1000.0) == 0.0


def _plane_depth(W, H, z_mm):
    return np.full((H, W), z_mm, np.int16)


def test_bilateral_filter_properties(pkg, oracle):
    W, H = 40, 30
    v = oracle.create_view(W, H)
    rgba = np.zeros((H, W, 4), np.uint8)
    # a constant plane stays constant in the interior; the border is upstream's zero floatImage border
    oracle.view_update(v, rgba, _plane_depth(W, H, 1500), bilateral=True)
    d = oracle.download_view_depth(v)
    assert np.all(d[2:-2, 2:-2] > 1.4999) and np.all(d[2:-2, 2:-2] < 1.5001)
    assert np.all(d[:2] == 0) and np.all(d[-2:] == 0) and np.all(d[:, :2] == 0) and np.all(d[:, -2:] == 0)
    # invalid pixels stay invalid and do not leak into neighbours
    mm = _plane_depth(W, H, 1500)
    mm[10:14, 10:14] = 0
    oracle.view_update(v, rgba, mm, bilateral=True)
    d = oracle.download_view_depth(v)
    assert np.all(d[10:14, 10:14] == -1.0)
    inner = d[4:-4, 4:-4]
    assert np.all((inner == -1.0) | (np.abs(inner - 1.5) < 1e-4))
    # noise is reduced; a depth step (0.5 m) is preserved
    rng = np.random.default_rng(0)
    mm = (1500 + rng.normal(0, 3, (H, W))).astype(np.int16)
    mm[:, W // 2:] += 500
    oracle.view_update(v, rgba, mm)
    raw = oracle.download_view_depth(v)
    oracle.view_update(v, rgba, mm, bilateral=True)
    d = oracle.download_view_depth(v)
    left_raw, left = raw[4:-4, 4:W // 2 - 4], d[4:-4, 4:W // 2 - 4]
    assert left.std() < 0.6 * left_raw.std()
    assert abs(d[4:-4, W // 2 + 4:-4].mean() - d[4:-4, 4:W // 2 - 4].mean() - 0.5) < 2e-3
    # the filter off path is unchanged by having used it
    oracle.view_update(v, rgba, mm)
    assert np.array_equal(oracle.download_view_depth(v), raw)


def test_bgr_to_rgba(oracle):
    W, H = 7, 5  # 35 pixels: not a multiple of four
    rng = np.random.default_rng(3)
    bgr = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    v = oracle.create_view(W, H)
    oracle.view_update_bgr(v, bgr, _plane_depth(W, H, 1000))
    out = oracle.download_view_rgba(v)
    assert np.array_equal(out[..., 0], bgr[..., 2]) and np.array_equal(out[..., 1], bgr[..., 1])
    assert np.array_equal(out[..., 2], bgr[..., 0]) and np.all(out[..., 3] == 255)


def test_depth_post_processing_known_cases(pkg, oracle):
    rows, cols = 24, 32
    # the reference pairs row with (cx, fx) and col with (cy, fy): choose cx inside the row range etc.
    intr = (30.0, 30.0, 11.5, 15.5)
    curr = np.full((rows, cols), 2000, np.int16)
    I = np.eye(4, dtype=np.float32)
    # identical frames, identity motion: every pixel is compared, none is blanked (row_u == row needs row >= 1)
    out, count = oracle.depth_post_processing(curr, curr, I, intr, 0.05, 0.0)
    assert np.array_equal(out, curr)
    assert count == (rows - 1) * (cols - 1)
    # previous keyframe 20 % farther: everything below filterArea * rows is blanked, rows above are kept
    prev = np.full((rows, cols), 2400, np.int16)
    out, count = oracle.depth_post_processing(curr, prev, I, intr, 0.05, 0.5)
    blank = np.zeros((rows, cols), bool)
    blank[13:, 1:] = True  # row > 0.5 * 24  and  row_u, col_v >= 1
    assert np.array_equal(out == 0, blank) and np.array_equal(out[~blank], curr[~blank])
    # below the threshold nothing happens; invalid current / previous pixels are skipped and not counted
    out, _ = oracle.depth_post_processing(curr, prev, I, intr, 0.25, 0.0)
    assert np.array_equal(out, curr)
    c2, p2 = curr.copy(), prev.copy()
    c2[5, 5] = 0; c2[6, 6] = -7; p2[8, 8] = 3
    out, count2 = oracle.depth_post_processing(c2, p2, I, intr, 0.05, 0.0)
    assert out[5, 5] == 0 and out[6, 6] == -7 and out[8, 8] == 2000 and count2 == (rows - 1) * (cols - 1) - 3
    # a translation along z by +0.4 m makes the 2.4 m keyframe consistent again
    T = I.copy(); T[2, 3] = 0.4
    out, _ = oracle.depth_post_processing(curr, prev, T, intr, 0.05, 0.0)
    # points move along the ray, so they reproject to the pixel (fx X / (z+0.4) + cx): still inside for most
    assert (out == 0).sum() == 0
