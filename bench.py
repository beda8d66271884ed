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
import gc
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


CPU_FRAMES = 96  # the CPU baseline runs on its own stretch of the sequence, whatever --steps says


def cpu_baseline(pkg, wl, params, frames, warm=20, budget_s=6.0):
    """CPU oracle (kind "port": the reference's CPU engine cannot be built here) on the same frames with the
    same call sequence, built as the reference builds its CPU code (-O3 -march=native, CMakeLists.txt:32,45) on THIS
    machine.  Parallelisation = upstream's CPU engine: OpenMP over visible blocks (integration) and pixels (raycast),
    the allocation pass sequential.  The first `warm` frames (cold map, cold caches) are not timed, like the GPU
    side's warm-up; then ~budget_s seconds each on 16 threads (the per-GPU share of the box), on every CPU this process
    may use, and on one thread -- of the same CPU_FRAMES-frame stretch, independent of --steps."""
    orc_pkg = ge.load_oracle()
    orc, flags = orc_pkg.open_native_oracle(pkg.CApi)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # the GPU box gives one GPU a 16-CPU share of the host (gpurun); `value` uses that share, all of it
    threads = max(1, min(orc.max_threads(), avail, int(os.environ.get("DSLAM_CPU_THREADS", "16"))))
    threads_all = max(1, min(orc.max_threads(), avail))
    rgba, depth, Ms = frames
    s = orc.create_scene(params)
    rs = orc.create_render_state(s, wl.W, wl.H)
    rs_free = orc.create_render_state(s, wl.W, wl.H)  # renderState_freeview, as in the GPU step
    v = orc.create_view(wl.W, wl.H)
    phases = {"update_view": 0.0, "allocate": 0.0, "integrate": 0.0, "raycast": 0.0}

    def run(first, budget, max_frames, split=None):
        n, t0 = 0, time.perf_counter()
        while n < max_frames and first + n < len(Ms):
            i = first + n
            ta = time.perf_counter()
            orc.view_update(v, rgba[i], depth[i], timestamp=float(i))
            tb = time.perf_counter()
            orc.allocate_scene_from_depth(s, v, rs, Ms[i], wl.intr)   # (ProcessFrame = these two calls + the ring push)
            tc = time.perf_counter()
            orc.integrate_into_scene(s, v, rs, Ms[i], wl.intr)
            td = time.perf_counter()
            orc.get_image(s, rs_free, Ms[i], wl.intr, pkg.IMAGE_DEPTH, download=True)
            te = time.perf_counter()
            if split is not None:
                for key, dt in zip(("update_view", "allocate", "integrate", "raycast"), (tb - ta, tc - tb, td - tc, te - td)):
                    split[key] += dt
            n += 1
            if budget is not None and time.perf_counter() - t0 > budget:
                break
        return n, time.perf_counter() - t0

    orc.set_threads(threads)
    warm = min(warm, max(0, len(Ms) - 12))
    run(0, None, warm)
    left = len(Ms) - warm
    n_16, dt_16 = run(warm, budget_s, max(1, left // 2), phases)
    out = {"value": n_16 / dt_16, "unit": "frames/s", "cores": threads, "kind": "port",
           "sample": (f"frames {warm}..{warm + n_16 - 1} of a {len(Ms)}-frame {wl.name} {wl.W}x{wl.H} sequence of its own "
                      f"(independent of --steps) after {warm} untimed warm-up frames, same call sequence (host images in, "
                      f"depth image out), {dt_16:.1f} s")}
    done = warm + n_16
    if threads_all != threads:
        orc.set_threads(threads_all)
        n_all, dt_all = run(done, budget_s, max(1, (len(Ms) - done) // 2))
        out["all_cpus_fps"] = n_all / dt_all
        out["all_cpus_threads"] = threads_all
        done += n_all
    orc.set_threads(1)
    n_1t, dt_1t = run(done, budget_s, max(1, len(Ms) - done))
    per = {k_: 1e3 * t / max(1, n_16) for k_, t in phases.items()}
    tot = sum(per.values())
    out.update({
        "one_thread_fps": n_1t / dt_1t,
        "one_thread_sample": f"frames {done}..{done + n_1t - 1}, {dt_1t:.1f} s",
        "ms_per_frame_by_call": {k_: round(x, 2) for k_, x in per.items()},
        "bounded_by": (f"{max(per, key=per.get)}: {max(per.values()):.1f} ms of {tot:.1f} ms per frame on {threads} threads "
                       f"(allocation, sequential in upstream's CPU engine: {per['allocate']:.1f} ms; integration "
                       f"{per['integrate']:.1f} ms and raycast {per['raycast']:.1f} ms are the OpenMP loops; compare the 1-thread "
                       f"and all-CPU figures: more cores move the rate little).  The GPU / CPU ratio says nothing about kernel "
                       f"quality, roofline.frac does"),
        "cores_available_to_process": avail, "cores_on_box": os.cpu_count(), "cpu_model": cpu_model(),
        "build": "g++ " + flags,
        "parallelisation": "OpenMP over visible blocks (integrate) and pixels (raycast); allocation sequential, as upstream's CPU engine"})
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script, one per GPU, BEFORE this process
    makes any GPU call (it never does), hand rank 0's JSON line through, fail if a rank fails."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE, text=True))
    outs = [p.communicate()[0] for p in procs]
    bad = [r for r, p in enumerate(procs) if p.returncode != 0]
    if args.spawn_dry_run:
        ranks = sorted((json.loads(line) for o in outs for line in o.splitlines() if line.startswith("{")),
                       key=lambda d: d["rank"])
        print(json.dumps({"spawn_dry_run": True, "n_gpus": args.gpus, "ranks": ranks, "failed_ranks": bad}), flush=True)
    else:
        for line in outs[0].splitlines():
            if line.startswith("{"):
                print(line, flush=True)
    if bad:
        raise SystemExit(f"bench.py: rank(s) {bad} failed")


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
    ap.add_argument("--spawn-dry-run", action="store_true",
                    help="with --gpus N: start the N ranks, let each report its rank / world size and exit before it touches a GPU")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args)   # (this process stays off the GPU; the ranks are children with RANK / LOCAL_RANK / WORLD_SIZE set)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.spawn_dry_run:
        print(json.dumps({"rank": rank, "local_rank": local_rank, "world_size": world}), flush=True)
        return

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
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    frames_all = generate_frames(args.workload, args.width, args.height, max(nframes, CPU_FRAMES) if want_cpu else nframes, workers)
    frames = tuple(x[:nframes] for x in frames_all)
    rgba_h, depth_h, Ms = frames

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libdslam_fusion has no CPU path")
    # Rehearsal of the N > 1 code on a ONE-GPU box (DSLAM_BENCH_REHEARSE=1; never the driver's command): every rank a process
    # on cuda:0, collectives over gloo on host tensors, the block exchange staged through the host (RCCL refuses two ranks on
    # one device).  The line it prints says so (`config.rehearsal`) and its rates mean nothing -- the ranks share the GPU; what
    # it shows is that the multi-rank path runs and that its self-checks (`map_checksum_equal`, `equals_unsharded_run_on_rank0`)
    # hold across real process boundaries.
    rehearse = os.environ.get("DSLAM_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    cdev = torch.device("cpu") if rehearse else torch.device("cuda", local_rank)   # where the small collectives' tensors live

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
        # (the driver script's own garbage collector: a full collection over the frame generator's objects took 30-50 ms in
        # the middle of the S-room loop -- as much as 200 frames' worth of time; what exists now is exempted from collections)
        gc.collect()
        gc.freeze()
        if timed_roofline:
            eng.kernel_timer_enable(True)
        barrier()
        t0 = time.perf_counter()
        trace = os.environ.get("DSLAM_BENCH_TRACE")   # diagnostics: which steps' calls keep the host for more than a millisecond?
        for i in range(Wm, Wm + K):
            if trace:
                ta = time.perf_counter()
            step(i)
            if trace and time.perf_counter() - ta > 1e-3:
                print(f"[bench trace] step {i}: {(time.perf_counter() - ta) * 1e3:.2f} ms on the host", file=sys.stderr)
        if trace:
            ta = time.perf_counter()
        eng.synchronize()  # drains the engine stream: every output image of the timed region is in host memory
        if trace:
            print(f"[bench trace] final synchronize: {(time.perf_counter() - ta) * 1e3:.2f} ms", file=sys.stderr)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        barrier()
        return t1 - t0, rs, rs_free

    elapsed, rs, rs_free = timed_loop(args.mode, True)
    if use_dist:
        t = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
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
    # `--reint` keyframes at corrected poses, blocks sharded over the ranks, one RCCL all-gather at the end.  Three forms of
    # the same batch, each undoing the one before it (old -> new, new -> old, old -> new):
    #   reference_calls   DeProcessFrame + ProcessFrame per keyframe, as DenseSlam::OnlineCorrection calls them
    #   stored_lists      the same loop, de-integrating from each keyframe's stored visible list (no allocation pass at the old pose)
    #   block_major       dslam_reintegrate_batch: allocation passes first, then every touched block loaded once (the headline)
    reint_out = None
    if args.reint > 0:
        try:
            import zlib
            from dslam_amd.harness import reintegrate as reint
            Kre = min(args.reint, K)
            ids = list(range(Wm + K - Kre, Wm + K))
            new_poses = []
            for n_, i in enumerate(ids):  # a smooth loop-closure correction: small rotation + translation drift
                T_new = wl.pose(i) @ synth.pose_matrix(synth.look_rotation(0.002 * (n_ + 1), 0.0), [0.01 * (n_ + 1), 0.0, 0.02])
                new_poses.append(synth.world_to_camera(T_new))
            old_poses = [Ms[i] for i in ids]
            chunk = 64

            def max_over_ranks(vals):
                t = torch.tensor(vals, device=cdev, dtype=torch.float64)
                if use_dist:
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                return [float(x) for x in t.tolist()]

            def checksum(sc, vox):
                """(voxel pool, hash table) as two integers; the pool is summed on the device."""
                eng.synchronize()
                torch.cuda.synchronize()
                pool = int(vox.view(torch.int64).sum().item())
                table = zlib.crc32(eng.download_hash_table(sc).tobytes())
                return pool, table

            def run_legs(sc, vw, rstate, vox, rank_, world_, collective):
                """The three forms on one map; returns their timings and the map checksum after each."""
                res, sums = {}, []
                make_ag = reint.make_staged_all_gather if rehearse else reint.make_torch_all_gather
                ag = make_ag(eng, sc, dist, eng.synchronize) if collective else None
                kw = dict(rank=rank_, world=world_, chunk_blocks=chunk, all_gather=ag, force_collective=collective)
                frames_dev = [("dev", rgba_d.data_ptr() + i * rgba_stride, depth_d.data_ptr() + i * depth_stride) for i in ids]
                tm = {}
                barrier()
                reint.reintegrate(eng, sc, vw, rstate, reint.Batch(frames_dev, old_poses, new_poses, wl.intr), timers=tm, **kw)
                barrier()
                res["reference_calls"] = tm
                sums.append(checksum(sc, vox))
                # keyframe images into a store; lists as a re-fusion at the corrected pose leaves them
                store = eng.create_frame_store(wl.W, wl.H, Kre)
                eng.frame_store_enable_lists(store, sc)
                eng.set_async(False)
                for n_, i in enumerate(ids):
                    eng.view_update_device(vw, rgba_d.data_ptr() + i * rgba_stride, depth_d.data_ptr() + i * depth_stride, timestamp=float(i))
                    eng.frame_store_put_view(store, n_, vw)
                    eng.allocate_scene_from_depth(sc, vw, rstate, new_poses[n_], wl.intr, only_update_visible_list=True)
                    eng.frame_store_put_visible_list(store, n_, sc, rstate)
                eng.set_async(args.mode != "sync")
                frames_st = [("store", store, n_) for n_ in range(Kre)]
                tm = {}
                barrier()
                reint.reintegrate(eng, sc, vw, rstate, reint.Batch(frames_st, new_poses, old_poses, wl.intr), timers=tm, stored_lists=True, **kw)
                barrier()
                res["stored_lists"] = tm
                sums.append(checksum(sc, vox))
                eng.reintegrate_batch(sc, vw, rstate, store, [], [], [], wl.intr)   # (set-up call: the scratch buffers)
                tm = {}
                barrier()
                reint.reintegrate(eng, sc, vw, rstate, reint.Batch(frames_st, old_poses, new_poses, wl.intr), timers=tm, batched=True, **kw)
                barrier()
                # the distinct blocks the batch loaded and its block-operations ((block, keyframe) pairs de-integrated or re-fused)
                tm["blocks_touched"], tm["block_operations"] = eng.reintegrate_batch_stats(sc)
                res["block_major"] = tm
                sums.append(checksum(sc, vox))
                store.close()
                return res, sums

            res, sums = run_legs(scene, view, rs, vox_t, rank, world, use_dist)
            reint_out = {"keyframes": Kre, "ranks_seen_by_rccl": dist.get_world_size() if use_dist else 1,
                         "scaling": "strong (fixed batch; allocation replicated on every rank, voxel blocks sharded, one all-gather "
                                    "of the blocks the batch touched)"}
            for form, tm in res.items():
                tot, rei, agt = max_over_ranks([tm["total_s"], tm["reintegrate_s"], tm["all_gather_s"]])
                reint_out[form] = {"keyframes_per_s": Kre / tot, "total_ms": tot * 1e3, "compute_ms": rei * 1e3, "all_gather_ms": agt * 1e3,
                                   "gathered_bytes": tm["gathered_bytes"], "dirty_blocks": tm["dirty_blocks"],
                                   **({"block_operations": int(tm["block_operations"]), "blocks_touched": int(tm["blocks_touched"])} if "block_operations" in tm else {}),
                                   "all_gather_GBps": (tm["gathered_bytes"] / agt / 1e9) if (use_dist and agt > 0) else None}
            reint_out["forms"] = ("reference_calls = DeProcessFrame + ProcessFrame per keyframe as DenseSlam.cpp:389-403 calls them; stored_lists / "
                                  "block_major de-integrate the blocks of the keyframe's own fusion-time list (not the reference's call sequence; "
                                  "like the whole path: parity unpinned, checked against the oracle only); block_major = dslam_reintegrate_batch, "
                                  "bit-identical to the stored_lists loop")
            reint_out["keyframes_per_s"] = reint_out["block_major"]["keyframes_per_s"]
            reint_out["all_gather_ms"] = reint_out["block_major"]["all_gather_ms"]
            reint_out["gathered_bytes"] = reint_out["block_major"]["gathered_bytes"]
            # (forms 1 and 3 need not leave the same bytes: DeProcessFrame as the reference calls it visits what an allocation
            # pass at the old pose finds today, the stored-list forms visit the keyframe's own blocks; parity of each form is
            # the test-suite's business, the checks below are about the ranks)
            if use_dist:
                # every rank must hold the same map after each exchange ...
                flat = [float(x % (1 << 52)) for pair in sums for x in pair]
                t_min = torch.tensor(flat, device=cdev, dtype=torch.float64)
                t_max = t_min.clone()
                dist.all_reduce(t_min, op=dist.ReduceOp.MIN)
                dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
                reint_out["map_checksum_equal"] = bool(torch.equal(t_min, t_max))
                # ... and rank 0 repeats everything UNSHARDED on a second map built from the same frames
                if rank == 0:
                    vox_ref = torch.empty(nlb * 512 * 8, dtype=torch.uint8, device=dev)
                    scene_ref = eng.create_scene(params, ext_voxel_blocks_dev=vox_ref.data_ptr())
                    view_ref = eng.create_view(wl.W, wl.H)
                    rs_ref = eng.create_render_state(scene_ref, wl.W, wl.H)
                    eng.set_async(False)
                    for i in range(Wm + K):
                        eng.view_update_device(view_ref, rgba_d.data_ptr() + i * rgba_stride, depth_d.data_ptr() + i * depth_stride, timestamp=float(i))
                        eng.process_frame(scene_ref, view_ref, rs_ref, Ms[i], wl.intr)
                    eng.set_async(args.mode != "sync")
                    saved = barrier
                    barrier = lambda: (eng.synchronize(), torch.cuda.synchronize())  # noqa: E731  (rank-local leg: no collective)
                    _, sums_ref = run_legs(scene_ref, view_ref, rs_ref, vox_ref, 0, 1, False)
                    barrier = saved
                    reint_out["equals_unsharded_run_on_rank0"] = bool(sums_ref == sums)
                    for o in (rs_ref, view_ref, scene_ref):
                        o.close()
        except Exception as ex:  # never lose the main line over the auxiliary measurement
            import traceback
            reint_out = {"error": repr(ex), "trace": traceback.format_exc()[-600:]}

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

    # ---- the same step through the drop-in: the C++ ITMLib mirror driven by the native driver harness (rank 0, N = 1 only)
    mirror_out = None
    if rank == 0 and world == 1 and not args.no_extra_rates:
        try:
            from dslam_amd.harness import mirror_bench
            eng.synchronize()
            m = mirror_bench.measure(keyframes=80, time_from=30, repeats=1, width=wl.W, height=wl.H, loops=["plain_raycast"], modes=("deferred",))
            row = m["plain_raycast"]["deferred"]
            mirror_out = {"fps": 1e6 / row["us_per_keyframe"], "us_per_keyframe": row["us_per_keyframe"],
                          "us_per_keyframe_without_image_fill": row["us_per_keyframe_without_image_fill"],
                          "host_us_in_calls": row["host_us_in_calls"],
                          "what": "UpdateView + IntegrateLocalMap + GetImage(FREECAMERA_DEPTH, read back) per keyframe through the C++ ITMLib mirror "
                                  "(itmlib/tests/driver_harness: the reference driver's calls, frames copied into the driver's own images, "
                                  "every call returning as the reference expects), 50 S-street keyframes"}
        except Exception as ex:
            mirror_out = {"error": repr(ex)}

    # ---- S-stress: the integrate kernel with every block of a 1 GiB pool visible (rank 0, N = 1 only)
    stress_out = None
    if rank == 0 and world == 1 and not args.no_stress:
        try:
            from dslam_amd.harness import stress
            eng.set_async(False)
            def stress_point(as_process_frame):
                r = stress.run(pkg, eng, n_side=64, iterations=20, W=wl.W, H=wl.H, as_process_frame=as_process_frame)
                return {"workload": r["workload"], "visible_blocks": r["visible_blocks"],
                        "achieved": r["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": r["achieved_GBps"] / HBM_PEAK_GBS, "avg_launch_us": r["ms_per_launch"] * 1e3,
                        "launches": r["iterations"], "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"]}
            stress_out = stress_point(True)            # the production launch: ProcessFrame's, ring push on
            stress_no_push = stress_point(False)       # IntegrateIntoScene's (what round 3 reported as `stress`)
        except Exception as ex:
            stress_out = {"error": repr(ex)}

    if rank == 0:
        # algorithmic bytes of the integrate kernel: per visible block 4 KiB read + 4 KiB write +
        # 16 B hash entry + 4 B list id; per launch the float depth image (4 B/px) and the RGBA image (4 B/px)
        alg_bytes = 8212.0 * blocks + 8.0 * wl.W * wl.H * launches
        avg_ms = int_ms / max(1, launches)
        achieved = (alg_bytes / max(1, launches)) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic, traffic_source = None, None
        for tag in ("r04", "r03", "r02", "r01"):  # PMC passes are separate rocprofv3 runs of this same command (profiles/)
            pmc_path = os.path.join(ROOT, "profiles", f"{tag}_integrate_pmc.json")
            if os.path.exists(pmc_path):
                traffic = json.load(open(pmc_path))["traffic_bytes_per_visible_block"] * blocks / max(1, launches)
                traffic_source = (f"profiles/{tag}_integrate_pmc.json: bytes per visible block from the builder's separate rocprofv3 "
                                  f"--pmc passes (FETCH_SIZE, WRITE_SIZE) of this command, scaled by this run's block count -- NOT "
                                  f"measured in this run")
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
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "avg_launch_us": avg_ms * 1e3, "launches": launches,
                         "algorithmic_bytes_per_launch": alg_bytes / max(1, launches)},
        }
        out["config"].update(extra)
        if rehearse:
            out["config"]["rehearsal"] = ("DSLAM_BENCH_REHEARSE=1: all ranks on cuda:0, gloo, host-staged exchange -- a check that the "
                                          "multi-rank path runs; the rates are NOT measurements")
        if mirror_out is not None:
            out["config"]["dropin_mirror_fps"] = mirror_out.get("fps")
            out["config"]["dropin_mirror"] = mirror_out
        if stress_out is not None:
            out["roofline"]["stress"] = stress_out
            if "error" not in stress_out:
                out["roofline"]["stress_no_push"] = stress_no_push
        if reint_out is not None:
            out["reintegration"] = reint_out
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(pkg, wl, params, frames_all)
            out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
