"""Known-answer tests for the CPU oracle.  The vectors were derived from the published
InfiniTAM v2 formulas, not from running reference code (the reference ships none: parity unpinned)."""
import ctypes as C

import numpy as np


def test_hash_index(oracle):
    f = oracle.lib.oracle_hash_index
    kat = {(0, 0, 0): 0, (1, 0, 0): 455773, (0, 1, 0): 475301, (0, 0, 1): 655287, (1, 2, 3): 363058,
           (-1, -1, -1): 505009, (-5, 7, 100): 531920, (32767, -32768, 12): 915255, (10, -3, 25): 413036}
    for (x, y, z), want in kat.items():
        assert f(x, y, z, 0x100000) == want


def test_hash_index_matches_numpy(oracle):
    rng = np.random.RandomState(0)
    b = rng.randint(-32768, 32768, size=(1000, 3)).astype(np.int64)
    want = ((b[:, 0] * 73856093) ^ (b[:, 1] * 19349669) ^ (b[:, 2] * 83492791)) & 0xFFFFF
    for row, w in zip(b, want):
        assert oracle.lib.oracle_hash_index(int(row[0]), int(row[1]), int(row[2]), 0x100000) == int(w)


def test_point_to_block(oracle):
    kat = {(0, 0, 0): ((0, 0, 0), 0), (7, 7, 7): ((0, 0, 0), 511), (8, 0, 0): ((1, 0, 0), 0), (-1, 0, 0): ((-1, 0, 0), 7),
           (-8, -9, 17): ((-1, -2, 2), 120), (-7, 15, -16): ((-1, 1, -2), 57)}
    out = (C.c_int * 3)()
    for p, (b, lin) in kat.items():
        assert oracle.lib.oracle_point_to_block(p[0], p[1], p[2], out) == lin
        assert tuple(out) == b


def test_single_voxel_sequence(oracle):
    sdf, w = C.c_int16(32767), C.c_uint8(0)
    f = oracle.lib.oracle_update_voxel_eta
    got = []
    for eta in (0.01, 0.005, -0.01, 0.02, 0.5):
        f(C.byref(sdf), C.byref(w), C.c_float(eta), C.c_float(0.02), 100)
        got.append((sdf.value, w.value))
    assert got == [(16383, 1), (12287, 2), (2730, 3), (10239, 4), (14744, 5)]


def test_reset_scene_state(pkg, oracle):
    p = pkg.SceneParams()
    s = oracle.create_scene(p)
    assert s.params.num_local_blocks == 0x40000 and s.n_entries == 1179648
    st = oracle.stats(s)
    assert st["last_free_block_id"] == 262143 and st["last_free_excess_id"] == 0x20000 - 1
    h = oracle.download_hash_table(s)
    assert h.nbytes == 18874368 and (h["ptr"] == -2).all() and (h["offset"] == 0).all()
    v = oracle.download_voxel_blocks(s, 0, 4)
    assert v.nbytes == 4 * 4096 and (v["sdf"] == 32767).all() and (v["w_depth"] == 0).all()
    assert (v.view(np.uint64) == 0x7FFF).all()
    a = oracle.download_allocation_list(s)
    assert np.array_equal(a, np.arange(0x40000, dtype=np.int32))
    s.close()


def test_matrix_inverse(oracle):
    rng = np.random.RandomState(1)
    for _ in range(20):
        A = np.eye(4, dtype=np.float32)
        q, _ = np.linalg.qr(rng.randn(3, 3))
        A[:3, :3] = q
        A[:3, 3] = rng.randn(3)
        m = np.ascontiguousarray(A.T).ravel()
        out = np.zeros(16, np.float32)
        assert oracle.lib.oracle_invert_matrix(m.ctypes.data_as(C.POINTER(C.c_float)),
                                               out.ctypes.data_as(C.POINTER(C.c_float))) == 0
        inv = out.reshape(4, 4).T
        assert np.allclose(inv @ A, np.eye(4), atol=1e-5)
