#!/usr/bin/env python3
"""A/B timing of libdslam_fusion.so builds on the bench scene (device-resident loop): per library, in a fresh process,
K frames of UpdateView -> ProcessFrame -> GetImage with the integrate kernel timed by packet-attached events.
usage: ab_libs.py [--rounds R] [--steps K] lib_a.so lib_b.so ...   (alternates the libraries R times)"""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(lib, K, Wm):
    sys.path.insert(0, ROOT)
    import numpy as np
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from dslam_amd.harness import synth
    import bench
    import time
    wl = synth.s_street(640, 480)
    n = K + Wm
    rgba_h, depth_h, Ms = bench.generate_frames("s_street", 640, 480, n, 16)
    import torch
    dev = torch.device("cuda", 0)
    rgba_d = torch.from_numpy(rgba_h).to(dev); depth_d = torch.from_numpy(depth_h).to(dev)
    torch.cuda.synchronize()
    pkg._share_torch_hip_runtime() if hasattr(pkg, "_share_torch_hip_runtime") else None
    eng = pkg.CApi(lib, "dslam_", has_engine_device=True, device=0)
    nlb = 0x40000
    while nlb < 9000 + 600 * n:
        nlb *= 2
    params = pkg.SceneParams(num_local_blocks=nlb, **wl.scene_kwargs)
    scene = eng.create_scene(params)
    view = eng.create_view(640, 480)
    rs = eng.create_render_state(scene, 640, 480); rs_free = eng.create_render_state(scene, 640, 480)
    eng.set_async(True)
    rs_, ds_ = 640 * 480 * 4, 640 * 480 * 2
    def step(i):
        eng.view_update_device(view, rgba_d.data_ptr() + i * rs_, depth_d.data_ptr() + i * ds_, timestamp=float(i))
        eng.process_frame(scene, view, rs, Ms[i], wl.intr)
        eng.get_image(scene, rs_free, Ms[i], wl.intr, pkg.IMAGE_DEPTH, download=False)
    for i in range(Wm):
        step(i)
    eng.synchronize()
    eng.kernel_timer_enable(True)
    t0 = time.perf_counter()
    for i in range(Wm, n):
        step(i)
    eng.synchronize()
    t1 = time.perf_counter()
    ms, launches, blocks = eng.kernel_timer_read()
    eng.kernel_timer_enable(False)
    eng.set_async(False)
    import zlib
    st = eng.stats(scene, rs)
    first = st["last_free_block_id"] + 1
    crc = zlib.crc32(eng.download_hash_table(scene).tobytes())
    for lo in range(first, nlb, 16384):
        crc = zlib.crc32(eng.download_voxel_blocks(scene, lo, min(16384, nlb - lo)).tobytes(), crc)
    crc = zlib.crc32(eng.download_last_seen(scene).tobytes(), crc)
    img = eng.get_image(scene, rs_free, Ms[n - 1], wl.intr, pkg.IMAGE_DEPTH)
    img_crc = zlib.crc32(np.ascontiguousarray(img).tobytes())
    print(json.dumps({"lib": os.path.basename(lib), "integrate_us": ms / launches * 1e3, "frame_us": (t1 - t0) / K * 1e6,
                      "blocks": blocks / launches, "map_crc": "%08x" % crc, "img_crc": "%08x" % img_crc}), flush=True)


if __name__ == "__main__":
    a = sys.argv[1:]
    if a and a[0] == "--child":
        child(a[1], int(a[2]), int(a[3]))
        sys.exit(0)
    rounds, K = 2, 150
    while a and a[0].startswith("--"):
        if a[0] == "--rounds": rounds = int(a[1])
        if a[0] == "--steps": K = int(a[1])
        a = a[2:]
    for r in range(rounds):
        for lib in a:
            res = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", os.path.abspath(lib), str(K), "20"],
                                 capture_output=True, text=True)
            line = [l for l in res.stdout.splitlines() if l.startswith("{")]
            print(line[-1] if line else "FAILED %s: %s" % (lib, res.stderr[-400:]), flush=True)
