"""Helper for GPU tests that need many synthetic frames: renders them on a worker pool in a process of its own (a test
process that has initialised the GPU must not fork) into three raw files next to `prefix`."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(prefix, n, workers, noise, outliers):
    import __graft_entry__ as ge
    ge.load_package()
    from dslam_amd.harness import memory_curves as mc
    from dslam_amd.harness import synth
    wl = synth.s_street(640, 480, stereo_noise_px=noise, outlier_frac=outliers)
    rgba = np.lib.format.open_memmap(prefix + "_rgba.npy", mode="w+", dtype=np.uint8, shape=(n, wl.H, wl.W, 4))
    depth = np.lib.format.open_memmap(prefix + "_depth.npy", mode="w+", dtype=np.int16, shape=(n, wl.H, wl.W))
    poses = np.lib.format.open_memmap(prefix + "_poses.npy", mode="w+", dtype=np.float32, shape=(n, 4, 4))
    step = 100
    for a in range(0, n, step):
        mc._WL = wl
        import multiprocessing as mp
        with mp.get_context("fork").Pool(workers) as pool:
            for i, (r, d, M) in zip(range(a, min(n, a + step)), pool.map(mc._frame, range(a, min(n, a + step)), chunksize=4)):
                rgba[i], depth[i], poses[i] = r, d, M
    for m in (rgba, depth, poses):
        m.flush()


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), float(sys.argv[5]))
