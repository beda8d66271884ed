#!/usr/bin/env python3
"""bench.py -- fusion + raycast frames/s of the MI355X TSDF engine (BASELINE.json metric).

A "step" is one pass of the hot path over one frame, exactly the call sequence DenseSlam::ProcessFrame times
as "Static map fusion" (reference DenseSlam.cpp:210-232) plus one free-camera depth raycast (DenseSlam.cpp:573-603):
    UpdateView (int16 mm -> float m)  ->  ProcessFrame (allocate + integrate)  ->  GetImage(FREECAMERA_DEPTH)
Inputs (synthetic KITTI-like S-street frames, 640x480) are resident in HBM before the timed region;
outputs stay on the device.  All calls go through the C ABI of libdslam_fusion.so.

N > 1 (torch.distributed.run, one rank per GPU): per-frame fusion does not shard so ranks are
independent replicas, each fusing its own map -> weak scaling, value = all frames / max rank time.

Prints ONE JSON line on rank 0.  `roofline` is the integrate kernel timed live with HIP events on the engine
stream; `cpu_baseline` is the CPU oracle (OpenMP, all host threads) on a bounded sample of the same frames.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable copy rate


def _gen_frame(args):
    name, W, H, i = args
    pkg = ge.load_package()
    from dslam_amd.harness import synth
    wl = getattr(synth, name)(W, H)
    return wl.frame(i)


def generate_frames(name, W, H, n, workers):
    """(rgba[n,H,W,4] u8, depth[n,H,W] i16, M[n,4,4] f32) -- closed-form ray casts, identical on every machine."""
    jobs = [(name, W, H, i) for i in range(n)]
    if workers > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(workers) as pool:
            res = pool.map(_gen_frame, jobs, chunksize=max(1, n // (workers * 4)))
    else:
        res = [_gen_frame(j) for j in jobs]
    return (np.stack([r[0] for r in res]), np.stack([r[1] for r in res]), np.stack([r[2] for r in res]))


def cpu_baseline(pkg, wl, params, frames, budget_s=20.0, max_frames=400):
    """CPU oracle (kind "port": the reference's CPU engine cannot be built here) on the first frames of
    the same workload, same call sequence, all host threads (OpenMP over visible blocks / pixels)."""
    orc_pkg = ge.load_oracle()
    orc = orc_pkg.open_oracle(pkg.CApi)
    # the GPU box gives one GPU a 16-CPU share (of 256 logical CPUs); use that share, all of it
    threads = max(1, min(orc.max_threads(), int(os.environ.get("DSLAM_CPU_THREADS", "16"))))
    orc.set_threads(threads)
    rgba, depth, Ms = frames
    s = orc.create_scene(params)
    rs = orc.create_render_state(s, wl.W, wl.H)
    rs_free = orc.create_render_state(s, wl.W, wl.H)  # renderState_freeview, as in the GPU step
    v = orc.create_view(wl.W, wl.H)
    n = 0
    t0 = time.perf_counter()
    while n < min(max_frames, len(Ms)):
        orc.view_update(v, rgba[n], depth[n], timestamp=float(n))
        orc.process_frame(s, v, rs, Ms[n], wl.intr)
        orc.get_image(s, rs_free, Ms[n], wl.intr, pkg.IMAGE_DEPTH, download=False)
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"frames 0..{n - 1} of the same {wl.name} 640x480 sequence, same call sequence, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="s_street")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--reint", type=int, default=32, help="keyframes in the sharded re-integration batch (0 = skip)")
    ap.add_argument("--host-io", action="store_true",
                    help="PCIe-inclusive variant: frames come from host buffers (dslam_view_update) and the raycast "
                         "depth image is downloaded every frame, as InfiniTamDriver does; never the headline value")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) and run the all-gather path even with one rank (plumbing check)")
    ap.add_argument("--sync", action="store_true", help="synchronous calls (reference driver behaviour) instead of pipelined")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    pkg = ge.load_package()
    from dslam_amd.harness import synth
    wl = getattr(synth, args.workload)(args.width, args.height)
    K, Wm = args.steps, args.warmup
    nframes = K + Wm
    # Synthetic frames first, while this process has not touched the GPU: the worker pool forks, and a forked child
    # of a GPU-initialised process is not safe.  Under a profiler that initialises the GPU before the program starts
    # (rocprofv3 --pmc preloads its tool library) there is no such moment, so no pool is used there.
    preloaded = os.environ.get("LD_PRELOAD", "") + os.environ.get("ROCP_TOOL_LIBRARIES", "")
    workers = 1 if "rocprof" in preloaded.lower() else max(1, min(16, (os.cpu_count() or 2) // max(1, world)))
    frames = generate_frames(args.workload, args.width, args.height, nframes, workers)
    rgba_h, depth_h, Ms = frames

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libdslam_fusion has no CPU path")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    # pools sized so the un-windowed map of the whole run fits (the reference's default 0x40000-block pool fills
    # after ~320 KITTI keyframes, memory.txt:320); 288 GB of HBM make a 4 GiB pool a non-issue
    need_blocks = 9000 + 600 * nframes
    nlb = 0x40000
    while nlb < need_blocks:
        nlb *= 2
    params = pkg.SceneParams(num_local_blocks=nlb, **wl.scene_kwargs)

    dev = torch.device("cuda", local_rank)
    rgba_d = torch.from_numpy(rgba_h).to(dev)
    depth_d = torch.from_numpy(depth_h).to(dev)
    torch.cuda.synchronize()

    eng = pkg.open_engine(local_rank)
    # the voxel-block array lives in a torch tensor so the re-integration all-gather (RCCL) can run on it in place
    vox_t = torch.empty(nlb * 512 * 8, dtype=torch.uint8, device=dev)
    scene = eng.create_scene(params, ext_voxel_blocks_dev=vox_t.data_ptr())
    rs = eng.create_render_state(scene, wl.W, wl.H)
    # ITMMainEngine::GetImage(FREECAMERA_*) raycasts through its own renderState_freeview, not the local map's
    # render state: the fusion visible list must survive from keyframe to keyframe
    rs_free = eng.create_render_state(scene, wl.W, wl.H)
    view = eng.create_view(wl.W, wl.H)
    eng.set_async(not (args.sync or args.host_io))  # --host-io: every call returns with its result, as InfiniTamDriver's do
    rgba_stride = wl.W * wl.H * 4
    depth_stride = wl.W * wl.H * 2

    if args.host_io:
        # the caller's frames and the image it reads back live in page-locked memory, as upstream's MemoryBlock keeps
        # every image that has a device side (dslam_host_alloc): uploads and the copy back are plain DMAs
        rgba_p, depth_p = eng.host_alloc(rgba_h.shape, np.uint8), eng.host_alloc(depth_h.shape, np.int16)
        rgba_p[...] = rgba_h
        depth_p[...] = depth_h
        rgba_h, depth_h = rgba_p, depth_p
        image_p = eng.host_alloc((wl.H, wl.W), np.float32)

    def step(i):
        if args.host_io:
            eng.view_update(view, rgba_h[i], depth_h[i], timestamp=float(i))
        else:
            eng.view_update_device(view, rgba_d.data_ptr() + i * rgba_stride, depth_d.data_ptr() + i * depth_stride,
                                   timestamp=float(i))
        eng.process_frame(scene, view, rs, Ms[i], wl.intr)
        eng.get_image(scene, rs_free, Ms[i], wl.intr, pkg.IMAGE_DEPTH, download=False, out=image_p if args.host_io else None)

    for i in range(Wm):
        step(i)
    eng.synchronize()
    eng.kernel_timer_enable(True)

    def barrier():
        if use_dist:  # (also with --force-dist on one rank: the RCCL barrier / all-reduce path is exercised)
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for i in range(Wm, Wm + K):
        step(i)
    eng.synchronize()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    barrier()
    elapsed = t1 - t0
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    int_ms, launches, blocks = eng.kernel_timer_read()
    eng.kernel_timer_enable(False)
    st = eng.stats(scene, rs)
    hits = int((eng.get_image(scene, rs_free, Ms[Wm + K - 1], wl.intr, pkg.IMAGE_DEPTH) > 0).sum())

    # ---- sharded global re-integration (BASELINE configs[4]): de-integrate + re-integrate the last
    # `--reint` keyframes at corrected poses, blocks sharded over the ranks, one RCCL all-gather at the end.
    reint_out = None
    if args.reint > 0:
        try:
            from dslam_amd.harness import reintegrate as reint
            Kre = min(args.reint, K)
            ids = list(range(Wm + K - Kre, Wm + K))
            new_poses = []
            for n_, i in enumerate(ids):  # a smooth loop-closure correction: small rotation + translation drift
                T_new = wl.pose(i) @ synth.pose_matrix(synth.look_rotation(0.002 * (n_ + 1), 0.0), [0.01 * (n_ + 1), 0.0, 0.02])
                new_poses.append(synth.world_to_camera(T_new))
            batch = reint.Batch([("dev", rgba_d.data_ptr() + i * rgba_stride, depth_d.data_ptr() + i * depth_stride) for i in ids],
                                [Ms[i] for i in ids], new_poses, wl.intr)
            chunk = 64
            ag = reint.make_torch_all_gather(vox_t, dist, chunk, eng.synchronize, api=eng, scene=scene) if use_dist else None
            timers = {}
            barrier()
            reint.reintegrate(eng, scene, view, rs, batch, rank=rank, world=world, chunk_blocks=chunk, all_gather=ag,
                              timers=timers, force_collective=use_dist)
            barrier()
            tt = torch.tensor([timers["total_s"], timers["reintegrate_s"], timers["all_gather_s"]], device=dev,
                              dtype=torch.float64)
            if use_dist:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tot, rei, agt = [float(x) for x in tt.tolist()]
            reint_out = {"keyframes": Kre, "keyframes_per_s": Kre / tot, "total_ms": tot * 1e3, "compute_ms": rei * 1e3,
                         "all_gather_ms": agt * 1e3, "gathered_bytes": timers["gathered_bytes"],
                         "all_gather_GBps": (timers["gathered_bytes"] / agt / 1e9) if (use_dist and agt > 0) else None,
                         "scaling": "strong (fixed batch; allocation replicated on every rank, voxel blocks sharded)"}
        except Exception as ex:  # never lose the main line over the auxiliary measurement
            reint_out = {"error": repr(ex)}

    out = None
    if rank == 0:
        # algorithmic bytes of the integrate kernel: per visible block 4 KiB read + 4 KiB write +
        # 16 B hash entry + 4 B list id; per launch the float depth image (4 B/px) and the RGBA image (4 B/px)
        alg_bytes = 8212.0 * blocks + 8.0 * wl.W * wl.H * launches
        avg_ms = int_ms / max(1, launches)
        achieved = (alg_bytes / max(1, launches)) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "r01_integrate_pmc.json")
        if os.path.exists(pmc_path):  # PMC passes are separate rocprofv3 runs of this same command (profiles/)
            traffic = json.load(open(pmc_path))["traffic_bytes_per_visible_block"] * blocks / max(1, launches)
        out = {
            "metric": "TSDF fusion+raycast frames/sec (640x480)",
            "value": world * K / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": K,
            "warmup": Wm,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 arithmetic on int16 TSDF / uint8 weight+colour voxels",
            "data": "synthetic",
            "config": {"workload": f"{wl.name} {wl.W}x{wl.H}: synthetic stand-in for KITTI 2011_09_30_drive_0033 stereo "
                                   f"(BASELINE configs[1]), fusion+raycast only, poses precomputed, voxel "
                                   f"{wl.scene_kwargs['voxel_size']} m, mu {wl.scene_kwargs['mu']} m, frustum "
                                   f"{wl.scene_kwargs['frustum_min']}-{wl.scene_kwargs['frustum_max']} m",
                       "calls": ("synchronous; page-locked host buffers in, depth image out over PCIe every frame" if args.host_io else
                                 "pipelined (async engine stream)" if not args.sync else "synchronous per call"),
                       "parallelism": "replicas" if world > 1 else "single GPU",
                       "voxel_block_pool": nlb,
                       "visible_blocks_per_frame": blocks / max(1, launches),
                       "allocated_blocks_end": nlb - 1 - st["last_free_block_id"],
                       "raycast_hits_last_frame": hits},
            "roofline": {"bound": "hbm", "kernel": "k_integrate<false>", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_launch_us": avg_ms * 1e3, "launches": launches,
                         "algorithmic_bytes_per_launch": alg_bytes / max(1, launches)},
        }
        if reint_out is not None:
            out["reintegration"] = reint_out
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(pkg, wl, params, frames)
            out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
