"""End-to-end accuracy of fusion + raycast on the synthetic sequences, in the reference's own terms: the 12 numbers
scripts/eval_raycast_depth.py prints (MAE / RMSE in mm, absrel, the delta thresholds), for the raycast depth
DenseSlam::SaveRaycastDepth would dump (uint16 = metres * 256, InfiniTamDriver.cpp:187-199) against the ground-truth
depth, next to the same numbers for the noisy input depth the map was fused from.  There is no dataset on the box
(and none in the container), so the ground truth is the analytic renderer's depth; input noise is the Kinect-style
sigma(z) of SURVEY.md 8d scaled to the scene.  Not a parity test -- parity is tests/; this says the reconstructed
surface is where it should be.

    python denseslam-global-consistency-h_amd/harness/quality.py [frames]   # one JSON line
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def noisy_mm(wl, i, rng, synth, sigma_scale):
    T_wc = wl.traj(i)
    prims = wl.prims
    if wl.cull_z is not None:
        cz = T_wc[2, 3]
        prims = [p for p in prims if p.hi[2] >= cz - wl.cull_z[0] and p.lo[2] <= cz + wl.cull_z[1]]
    z, rgba = synth.render(prims, wl.intr.astype(np.float64), wl.W, wl.H, T_wc)
    sigma = sigma_scale * (0.0012 + 0.0019 * (np.where(np.isfinite(z), z, 0.0) * wl.noise_z_scale - 0.4) ** 2)
    zn = z + rng.standard_normal(z.shape) * sigma
    mm = synth.depth_to_mm_kitti(zn, wl.max_depth or 40.0) if wl.kitti_depth else synth.depth_to_mm_rgbd(zn, wl.max_depth)
    return z, rgba, mm, synth.world_to_camera(T_wc)


def run(workload, frames, eng, pkg, synth, evalio):
    wl = getattr(synth, workload)(640, 480)
    # the sigma(z) law is a Kinect's (metres, 0.4-4 m); the street scene spans 40 m, so z is mapped onto that range
    wl.noise_z_scale = 0.1 if workload == "s_street" else 1.0
    sigma_scale = 10.0 if workload == "s_street" else 1.0
    rng = np.random.default_rng(1234)
    p = pkg.SceneParams(**wl.scene_kwargs)
    s = eng.create_scene(p)
    rs, rsf = eng.create_render_state(s, wl.W, wl.H), eng.create_render_state(s, wl.W, wl.H)
    v = eng.create_view(wl.W, wl.H)
    acc = {"raycast": [], "input": []}
    for i in range(frames):
        z, rgba, mm, M = noisy_mm(wl, i, rng, synth, sigma_scale)
        eng.view_update(v, rgba, mm, timestamp=float(i))
        eng.process_frame(s, v, rs, M, wl.intr)
        if i < 5:
            continue  # let the map see the surface a few times first
        png = eng.get_depth_image_int16(s, rsf, M, wl.intr, 256)  # the dump format: (int16)(metres * 256)
        gt = np.where(np.isfinite(z), z, 0.0)
        lim = wl.scene_kwargs["frustum_max"]
        gt = np.where(gt < lim, gt, 0.0)  # the map holds nothing beyond the view frustum
        acc["raycast"].append(evalio.depth_metrics(evalio.png16_to_depth(png), gt))
        acc["input"].append(evalio.depth_metrics(mm.astype(np.float64) / 1000.0, gt))
    out = {"workload": f"{wl.name} 640x480", "frames": frames, "voxel_size_m": wl.scene_kwargs["voxel_size"],
           "mu_m": wl.scene_kwargs["mu"]}
    for k, rows in acc.items():
        keys = [q for q in rows[0] if q != "mask_number"]
        out[k] = {q: float(np.mean([r[q] for r in rows])) for q in keys}
        out[k]["mask_number_mean"] = float(np.mean([r["mask_number"] for r in rows]))
    return out


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    pkg = ge.load_package()
    from dslam_amd.harness import evalio, synth
    eng = pkg.open_engine(0)
    res = {"metric": "eval_raycast_depth numbers (errors in mm) of raycast depth and of the fused-from input depth vs ground truth",
           "runs": [run(w, frames, eng, pkg, synth, evalio) for w in ("s_room", "s_street")]}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
