#!/usr/bin/env python3
"""bench.py -- fusion + raycast frames/s of the MI355X TSDF engine (BASELINE.json metric).

A "step" is one pass of the hot path over one frame, exactly the call sequence DenseSlam::ProcessFrame times as
"Static map fusion" (reference DenseSlam.cpp:210-232) plus one free-camera depth raycast (DenseSlam.cpp:573-603):
    UpdateView (H2D, int16 mm -> float m)  ->  ProcessFrame (allocate + integrate)  ->  GetImage(FREECAMERA_DEPTH) (D2H)
on synthetic KITTI-like S-street frames, 640x480, through the C ABI of libdslam_fusion.so.

`value` is the PCIe-INCLUSIVE rate of that sequence: every frame starts in page-locked host memory
and its raycast depth image ends in page-locked host memory, pipelined the MI355X way -- frame i + 1 crosses PCIe on
the engine's copy stream while frame i's kernels run, the render kernel stores the image straight into host memory, a
fence per frame tells the consumer when.  Two more rates of the same frames are reported in `config`:
`device_resident_fps` (inputs already in HBM, outputs left there: the kernels alone) and `synchronous_fps` (every call
returns with its result, as the reference's InfiniTamDriver calls do; same page-locked buffers).

N > 1 (torch.distributed.run, one rank per GPU): per-frame fusion does not shard so ranks are
independent replicas, each fusing its own map -> weak scaling, value = all frames / max rank time.

Prints ONE JSON line on rank 0.  `roofline` = the integrate kernel of the timed region, timed live with HIP events
attached to its dispatch packets on the engine stream, plus (`roofline.stress`) the same kernel on the S-stress map
(V = 262,144 blocks: 2.15 GB per launch, past the 256 MiB Infinity Cache).  `cpu_baseline` = the CPU oracle built
-O3 -march=native on this box, OpenMP over visible blocks / pixels like upstream's CPU engine, warm-up excluded.
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
OUT_RING = 3           # page-locked output images in flight


def _gen_frame(args):
    name, W, H, i = args
    ge.load_package()
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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(pkg, wl, params, frames, warm=20, budget_s=12.0, budget_1t_s=8.0):
    """CPU oracle (kind "port": the reference's CPU engine cannot be built here) on the same frames with the
    same call sequence, built as the reference builds its CPU code (-O3 -march=native, CMakeLists.txt:32,45) on THIS
    machine.  Parallelisation = upstream's CPU engine: OpenMP over visible blocks (integration) and pixels (raycast),
    the allocation pass sequential.  The first `warm` frames (cold map, cold caches) are not timed, like the GPU
    side's warm-up; then ~budget_s seconds on all the threads this process may use, then ~budget_1t_s on one."""
    orc_pkg = ge.load_oracle()
    orc, flags = orc_pkg.open_native_oracle(pkg.CApi)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # the GPU box gives one GPU a 16-CPU share of the host (gpurun); use that share, all of it, unless told otherwise
    threads = max(1, min(orc.max_threads(), avail, int(os.environ.get("DSLAM_CPU_THREADS", "16"))))
    rgba, depth, Ms = frames
    s = orc.create_scene(params)
    rs = orc.create_render_state(s, wl.W, wl.H)
    rs_free = orc.create_render_state(s, wl.W, wl.H)  # renderState_freeview, as in the GPU step
    v = orc.create_view(wl.W, wl.H)

    def run(first, budget, max_frames):
        n, t0 = 0, time.perf_counter()
        while n < max_frames and first + n < len(Ms):
            i = first + n
            orc.view_update(v, rgba[i], depth[i], timestamp=float(i))
            orc.process_frame(s, v, rs, Ms[i], wl.intr)
            orc.get_image(s, rs_free, Ms[i], wl.intr, pkg.IMAGE_DEPTH, download=True)
            n += 1
            if budget is not None and time.perf_counter() - t0 > budget:
                break
        return n, time.perf_counter() - t0

    orc.set_threads(threads)
    warm = min(warm, max(0, len(Ms) - 8))
    run(0, None, warm)
    n_all, dt_all = run(warm, budget_s, len(Ms) - warm - 4)
    orc.set_threads(1)
    n_1t, dt_1t = run(warm + n_all, budget_1t_s, max(1, len(Ms) - warm - n_all))
    return {"value": n_all / dt_all, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": (f"frames {warm}..{warm + n_all - 1} of the same {wl.name} {wl.W}x{wl.H} sequence after {warm} untimed "
                       f"warm-up frames, same call sequence (host images in, depth image out), {dt_all:.1f} s"),
            "one_thread_fps": n_1t / dt_1t,
            "one_thread_sample": f"frames {warm + n_all}..{warm + n_all + n_1t - 1}, {dt_1t:.1f} s",
            "cores_available_to_process": avail, "cores_on_box": os.cpu_count(), "cpu_model": cpu_model(),
            "build": "g++ " + flags,
            "parallelisation": "OpenMP over visible blocks (integrate) and pixels (raycast); allocation sequential, as upstream's CPU engine"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="s_street")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stress", action="store_true", help="skip the S-stress roofline launch (roofline.stress)")
    ap.add_argument("--no-extra-rates", action="store_true", help="skip the device-resident and synchronous loops")
    ap.add_argument("--reint", type=int, default=32, help="keyframes in the sharded re-integration batch (0 = skip)")
    ap.add_argument("--mode", default="pipelined", choices=["pipelined", "device", "sync"],
                    help="what `value` measures: pipelined = PCIe-inclusive, overlapped (default; the contract sequence); "
                         "device = inputs resident in HBM, outputs left there; sync = PCIe-inclusive, every call synchronous")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) and run the all-gather path even with one rank (plumbing check)")
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
    view = eng.create_view(wl.W, wl.H)
    rgba_stride = wl.W * wl.H * 4
    depth_stride = wl.W * wl.H * 2

    # the caller's frames and the images it reads back live in page-locked memory, as upstream's MemoryBlock keeps every
    # image that has a device side (dslam_host_alloc): uploads and read-backs are DMAs / direct stores
    # (one record per frame: the RGBA image, then its int16 depth image -- such a frame goes up as one copy)
    rec = eng.host_alloc((nframes, rgba_stride + depth_stride), np.uint8)
    rgba_p = rec[:, :rgba_stride].reshape(nframes, wl.H, wl.W, 4)
    depth_p = rec[:, rgba_stride:].view(np.int16).reshape(nframes, wl.H, wl.W)
    rgba_p[...] = rgba_h
    depth_p[...] = depth_h
    image_p = eng.host_alloc((OUT_RING, wl.H, wl.W), np.float32)
    fences = [eng.fence_create() for _ in range(OUT_RING)]

    def barrier():
        if use_dist:  # (also with --force-dist on one rank: the RCCL barrier / all-reduce path is exercised)
            dist.barrier()
        torch.cuda.synchronize()

    def timed_loop(mode, timed_roofline):
        """Wm untimed + K timed steps of `mode` on a freshly reset map; returns (seconds, render states)."""
        eng.set_async(False)
        eng.reset_scene(scene)
        rs = eng.create_render_state(scene, wl.W, wl.H)
        # ITMMainEngine::GetImage(FREECAMERA_*) raycasts through its own renderState_freeview, not the local map's
        # render state: the fusion visible list must survive from keyframe to keyframe
        rs_free = eng.create_render_state(scene, wl.W, wl.H)
        eng.set_async(mode != "sync")

        def step(i):
            if mode == "device":
                eng.view_update_device(view, rgba_d.data_ptr() + i * rgba_stride, depth_d.data_ptr() + i * depth_stride,
                                       timestamp=float(i))
                eng.process_frame(scene, view, rs, Ms[i], wl.intr)
                eng.get_image(scene, rs_free, Ms[i], wl.intr, pkg.IMAGE_DEPTH, download=False)
                return
            slot = i % OUT_RING
            if mode == "pipelined":
                eng.fence_wait(fences[slot])  # image i - OUT_RING has landed (its consumer may take it) before slot is reused
            eng.view_update(view, rgba_p[i], depth_p[i], timestamp=float(i))
            eng.process_frame(scene, view, rs, Ms[i], wl.intr)
            eng.get_image(scene, rs_free, Ms[i], wl.intr, pkg.IMAGE_DEPTH, out=image_p[slot])
            if mode == "pipelined":
                eng.fence_record(fences[slot])

        for i in range(Wm):
            step(i)
        eng.synchronize()
        if timed_roofline:
            eng.kernel_timer_enable(True)
        barrier()
        t0 = time.perf_counter()
        for i in range(Wm, Wm + K):
            step(i)
        eng.synchronize()  # drains the engine stream: every output image of the timed region is in host memory
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        barrier()
        return t1 - t0, rs, rs_free

    elapsed, rs, rs_free = timed_loop(args.mode, True)
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    int_ms, launches, blocks = eng.kernel_timer_read()
    eng.kernel_timer_enable(False)
    eng.set_async(False)
    st = eng.stats(scene, rs)
    last_image = np.array(image_p[(Wm + K - 1) % OUT_RING]) if args.mode != "device" else \
        eng.get_image(scene, rs_free, Ms[Wm + K - 1], wl.intr, pkg.IMAGE_DEPTH)
    hits = int((last_image > 0).sum())
    eng.set_async(args.mode != "sync")

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
            ag = reint.make_torch_all_gather(eng, scene, dist, eng.synchronize) if use_dist else None
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
                         "dirty_blocks": timers["dirty_blocks"],
                         "all_gather_GBps": (timers["gathered_bytes"] / agt / 1e9) if (use_dist and agt > 0) else None,
                         "scaling": "strong (fixed batch; allocation replicated on every rank, voxel blocks sharded, "
                                    "one all-gather of the blocks the batch touched)"}
            # the same batch undone again (poses back to the originals), this time de-integrating from each keyframe's
            # stored visible list instead of an allocation pass at the old pose (dslam_deprocess_frame_stored)
            store = eng.create_frame_store(wl.W, wl.H, Kre)
            eng.frame_store_enable_lists(store, scene)
            eng.set_async(False)
            for n_, i in enumerate(ids):  # keyframe images into the store; lists as a re-fusion at the corrected pose leaves them
                eng.view_update_device(view, rgba_d.data_ptr() + i * rgba_stride, depth_d.data_ptr() + i * depth_stride, timestamp=float(i))
                eng.frame_store_put_view(store, n_, view)
                eng.allocate_scene_from_depth(scene, view, rs, new_poses[n_], wl.intr, only_update_visible_list=True)
                eng.frame_store_put_visible_list(store, n_, scene, rs)
            batch2 = reint.Batch([("store", store, n_) for n_ in range(Kre)], new_poses, [Ms[i] for i in ids], wl.intr)
            eng.set_async(args.mode != "sync")
            timers2 = {}
            barrier()
            reint.reintegrate(eng, scene, view, rs, batch2, rank=rank, world=world, chunk_blocks=chunk, all_gather=ag,
                              timers=timers2, force_collective=use_dist, stored_lists=True)
            barrier()
            t2 = torch.tensor([timers2["total_s"]], device=dev, dtype=torch.float64)
            if use_dist:
                dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            reint_out["stored_lists_keyframes_per_s"] = Kre / float(t2.item())
            reint_out["stored_lists_total_ms"] = float(t2.item()) * 1e3
            store.close()
        except Exception as ex:  # never lose the main line over the auxiliary measurement
            reint_out = {"error": repr(ex)}

    # ---- the same frames at the two other call disciplines (rank-local; not `value`)
    extra = {}
    if not args.no_extra_rates:
        for mode, key in (("device", "device_resident_fps"), ("sync", "synchronous_fps"), ("pipelined", "pipelined_fps")):
            if mode == args.mode:
                continue
            try:
                dt, rs_x, rs_free_x = timed_loop(mode, False)
                extra[key] = K / dt
                rs_x.close(); rs_free_x.close()
            except Exception as ex:
                extra[key] = repr(ex)
        eng.set_async(False)

    # ---- S-stress: the integrate kernel with every block of a 1 GiB pool visible (rank 0, N = 1 only)
    stress_out = None
    if rank == 0 and world == 1 and not args.no_stress:
        try:
            from dslam_amd.harness import stress
            eng.set_async(False)
            r = stress.run(pkg, eng, n_side=64, iterations=20, W=wl.W, H=wl.H)
            stress_out = {"workload": r["workload"], "visible_blocks": r["visible_blocks"],
                          "achieved": r["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": r["achieved_GBps"] / HBM_PEAK_GBS, "avg_launch_us": r["ms_per_launch"] * 1e3,
                          "launches": r["iterations"], "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"]}
        except Exception as ex:
            stress_out = {"error": repr(ex)}

    if rank == 0:
        # algorithmic bytes of the integrate kernel: per visible block 4 KiB read + 4 KiB write +
        # 16 B hash entry + 4 B list id; per launch the float depth image (4 B/px) and the RGBA image (4 B/px)
        alg_bytes = 8212.0 * blocks + 8.0 * wl.W * wl.H * launches
        avg_ms = int_ms / max(1, launches)
        achieved = (alg_bytes / max(1, launches)) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        for tag in ("r02", "r01"):  # PMC passes are separate rocprofv3 runs of this same command (profiles/)
            pmc_path = os.path.join(ROOT, "profiles", f"{tag}_integrate_pmc.json")
            if os.path.exists(pmc_path):
                traffic = json.load(open(pmc_path))["traffic_bytes_per_visible_block"] * blocks / max(1, launches)
                break
        calls = {"pipelined": "PCIe-inclusive, pipelined: page-locked host frames in (copy stream, overlapped with the previous "
                              "frame's kernels), raycast depth image stored into page-locked host memory every frame, one fence per frame",
                 "device": "inputs resident in HBM, outputs left on the device, calls pipelined on the engine stream",
                 "sync": "PCIe-inclusive, every call synchronous like InfiniTamDriver's"}[args.mode]
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
                       "calls": calls,
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
        out["config"].update(extra)
        if stress_out is not None:
            out["roofline"]["stress"] = stress_out
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
