#!/usr/bin/env python3
"""The fusion kernel as dslam_process_frame launches it (visible-list ring push on) on the S-stress map (V = 262,144: every
block of a 1 GiB pool visible), per library build (push_variants.sh), each in a fresh process, alternated twice; kernel time
by the events attached to the dispatch packet.  Prints one JSON line per run and a summary."""
import json, os, subprocess, sys, zlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(lib):
    sys.path.insert(0, ROOT)
    import numpy as np
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from dslam_amd.harness import stress
    eng = pkg.CApi(lib, "dslam_", has_engine_device=True, device=0)
    W, H, n_side = 640, 480, 64
    n = n_side ** 3
    params = pkg.SceneParams(voxel_size=0.01, mu=0.04, max_w=100, frustum_min=0.2, frustum_max=20.0,
                             num_local_blocks=n, num_buckets=0x100000, num_excess=0x20000)
    scene = eng.create_scene(params)
    rs, view = eng.create_render_state(scene, W, H), eng.create_view(W, H)
    table, visible, excess_list, last_free_ex = stress.build_lattice_state(pkg, n_side, params.num_buckets, params.num_excess)
    eng.upload_scene_state(scene, hash_table=table, allocation_list=np.arange(n, dtype=np.int32), last_free_block_id=-1,
                           excess_list=excess_list, last_free_excess_id=last_free_ex)
    eng.upload_visible_ids(rs, visible)
    eng.view_update(view, np.full((H, W, 4), 128, np.uint8), np.full((H, W), 30000, np.int16))
    M = np.eye(4, dtype=np.float32)
    intr = np.array([100.0, 100.0, (W - 1) / 2.0, (H - 1) / 2.0], np.float32)
    out = {"lib": os.path.basename(lib)}
    for name, call in (("process_frame", lambda: eng.process_frame(scene, view, rs, M, intr)),
                       ("integrate_into_scene", lambda: eng.integrate_into_scene(scene, view, rs, M, intr))):
        call(); call()
        eng.synchronize()
        eng.kernel_timer_enable(True)
        for _ in range(20):
            call()
        ms, launches, blocks = eng.kernel_timer_read()
        eng.kernel_timer_enable(False)
        us = ms / launches * 1e3
        out[name + "_us"] = round(us, 1)
        out[name + "_frac"] = round((8212.0 * blocks / launches + 8.0 * W * H) / (us * 1e-6) / 8e12, 4)
        out["visible"] = blocks // launches
    crc = zlib.crc32(eng.download_last_seen(scene).tobytes())
    crc = zlib.crc32(eng.download_voxel_blocks(scene, 0, 4096).tobytes(), crc)
    out["crc_last_seen_and_4096_blocks"] = "%08x" % crc
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    a = sys.argv[1:]
    if a and a[0] == "--child":
        child(a[1])
        sys.exit(0)
    rows = []
    for rnd in range(2):
        for lib in a:
            res = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib], capture_output=True, text=True, timeout=300)
            line = [l for l in res.stdout.splitlines() if l.startswith("{")]
            if res.returncode != 0 or not line:
                print("FAILED", lib, res.stderr[-400:], flush=True)
                continue
            print(line[-1], flush=True)
            rows.append(json.loads(line[-1]))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "push_variants.json"), "w"), indent=1)
