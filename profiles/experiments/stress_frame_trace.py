"""Per-dispatch kernel times of ProcessFrame / DeProcessFrame on the S-stress map (all 262,144 blocks visible).
Run under `rocprofv3 --kernel-trace --output-format csv`; profiles/experiments/stress_frame_trace.sh prints the dispatches."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "denseslam-global-consistency-h_amd", "harness"))
import maint_bench as mb


def main():
    pkg = mb.ge.load_package()
    from dslam_amd.harness import stress
    eng = pkg.open_engine(0)
    W, H = 640, 480
    view = eng.create_view(W, H)
    far = np.full((H, W), 30000, dtype=np.int16)
    rgba = np.full((H, W, 4), 128, dtype=np.uint8)
    M = np.eye(4, dtype=np.float32)
    intr = np.array([100.0, 100.0, (W - 1) / 2.0, (H - 1) / 2.0], np.float32)
    scene, rs, n, visible = mb.lattice_scene(pkg, eng, stress, 64, False, W, H)
    eng.upload_visible_ids(rs, visible)
    eng.view_update(view, rgba, far)
    for _ in range(4):
        eng.process_frame(scene, view, rs, M, intr)
        eng.synchronize()
    for _ in range(3):
        eng.deprocess_frame(scene, view, rs, M, intr)
        eng.synchronize()
    print("visible", eng.stats(scene, rs)["no_visible_entries"])


main()
