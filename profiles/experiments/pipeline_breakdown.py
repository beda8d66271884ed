#!/usr/bin/env python3
"""Where the PCIe-inclusive pipelined loop of bench.py spends its time: the four combinations of {frames resident in
HBM, frames uploaded from page-locked host memory on the copy stream} x {depth image left on the device, depth image
stored into page-locked host memory by the render kernel}, each with the host's enqueue time (loop without the final
wait) next to the total.  usage: python profiles/experiments/pipeline_breakdown.py [steps]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    Wm = 10
    pkg = ge.load_package()
    from dslam_amd.harness import synth
    wl = synth.s_street(640, 480)
    rgba_h, depth_h, Ms = bench.generate_frames("s_street", 640, 480, K + Wm, 16)
    import torch
    dev = torch.device("cuda", 0)
    rgba_d, depth_d = torch.from_numpy(rgba_h).to(dev), torch.from_numpy(depth_h).to(dev)
    cs_, ds_ = wl.W * wl.H * 4, wl.W * wl.H * 2
    eng = pkg.open_engine(0)
    params = pkg.SceneParams(num_local_blocks=0x40000, **wl.scene_kwargs)
    scene = eng.create_scene(params)
    view = eng.create_view(wl.W, wl.H)
    rec = eng.host_alloc((K + Wm, cs_ + ds_), np.uint8)  # one record per frame: RGBA image, then the int16 depth image
    rgba_p = rec[:, :cs_].reshape(K + Wm, wl.H, wl.W, 4)
    depth_p = rec[:, cs_:].view(np.int16).reshape(K + Wm, wl.H, wl.W)
    rgba_p[...] = rgba_h
    depth_p[...] = depth_h
    R = 3
    image_p = eng.host_alloc((R, wl.H, wl.W), np.float32)
    fences = [eng.fence_create() for _ in range(R)]
    cs, ds = wl.W * wl.H * 4, wl.W * wl.H * 2
    out = {}
    for name, host_in, host_out in (("dev_in/dev_out", 0, 0), ("dev_in/dev_out + 1 event record per frame", 0, 2), ("dev_in/dev_out + 2 event records per frame", 0, 3),
                                    ("host_in/dev_out", 1, 0), ("dev_in/host_out", 0, 1), ("host_in/host_out", 1, 1)):
        eng.set_async(False)
        eng.reset_scene(scene)
        rs, free = eng.create_render_state(scene, wl.W, wl.H), eng.create_render_state(scene, wl.W, wl.H)
        eng.set_async(True)

        def step(i):
            slot = i % R
            if host_out == 1:
                eng.fence_wait(fences[slot])
            if host_in:
                eng.view_update(view, rgba_p[i], depth_p[i], timestamp=float(i))
            else:
                eng.view_update_device(view, rgba_d.data_ptr() + i * cs, depth_d.data_ptr() + i * ds, timestamp=float(i))
            eng.process_frame(scene, view, rs, Ms[i], wl.intr)
            if host_out == 1:
                eng.get_image(scene, free, Ms[i], wl.intr, pkg.IMAGE_DEPTH, out=image_p[slot])
                eng.fence_record(fences[slot])
            else:
                eng.get_image(scene, free, Ms[i], wl.intr, pkg.IMAGE_DEPTH, download=False)
                for k in range(host_out - 1 if host_out > 1 else 0):  # (what an event record costs the stream, by itself)
                    eng.fence_record(fences[(slot + k) % R])

        for i in range(Wm):
            step(i)
        eng.synchronize()
        t0 = time.perf_counter()
        for i in range(Wm, Wm + K):
            step(i)
        t1 = time.perf_counter()
        eng.synchronize()
        t2 = time.perf_counter()
        out[name] = {"enqueue_us_per_step": (t1 - t0) / K * 1e6, "total_us_per_step": (t2 - t0) / K * 1e6}
        rs.close(); free.close()
    eng.set_async(False)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
