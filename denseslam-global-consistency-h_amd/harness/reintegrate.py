"""Sharded global re-integration after a loop closure (BASELINE config 4).

DenseSlam::OnlineCorrection (reference DenseSlam.cpp:298-432) de-integrates every corrected keyframe at its old pose
and re-integrates it at the new one.  A voxel's new value depends only on its own old value and the frame sequence,
so voxel blocks are independent units and the batch shards by voxel-block slot:

  1. every rank holds the full map replica and the same list of (frame, old pose, new pose);
  2. every rank runs the allocation passes of all frames -- deterministic and bit-identical, so hash tables and
     free lists stay identical on all ranks without any exchange;
  3. rank g de-integrates / re-integrates only blocks whose slot chunk (slot // chunk_blocks) % world == g
     (dslam_scene_set_shard; round-robin chunks balance the load), in keyframe order -- so every voxel sees exactly the
     update sequence of the single-GPU run;
  4. ONE collective at the end, over exactly the blocks the batch touched: the (de-)integration kernels mark every
     visible resident block they walk over (before their shard test, so all ranks hold the same marks); every rank
     derives the same per-shard lists of dirty slots, packs its own shard's blocks in list order,
     all_gather_into_tensor (padded to the longest list; RCCL over xGMI on GPUs) moves them, and the other shards'
     blocks are put in place from the same lists.  No ids travel, no counts are exchanged, and nothing is assumed about
     WHERE in the pool live blocks sit (after decay / sliding window / swapping the used slots are not a top range).

The result is byte-identical to the unsharded run (tests/test_multigpu_gloo.py).  The module is engine-agnostic:
`api` is a bound C ABI (`CApi`), either the HIP library (collective = NCCL/RCCL on torch CUDA tensors) or -- for the
multi-rank CPU tests -- the oracle (collective = gloo on numpy buffers).
"""
import time

import numpy as np

BLOCK_BYTES = 512 * 8


class Batch:
    """The corrected keyframes: frames (("dev", rgba_ptr, depth_ptr), ("host", rgba, depth_mm) or ("store", frame_store,
    slot)) plus old / new world->camera poses and the intrinsics."""

    def __init__(self, frames, old_poses, new_poses, intr):
        self.frames, self.old_poses, self.new_poses, self.intr = frames, old_poses, new_poses, intr

    def __len__(self):
        return len(self.old_poses)


def _update_view(api, view, frame, ts):
    if frame[0] == "dev":
        api.view_update_device(view, frame[1], frame[2], timestamp=ts)
    elif frame[0] == "store":
        api.view_update_from_store(view, frame[1], frame[2], timestamp=ts)
    else:
        api.view_update(view, frame[1], frame[2], timestamp=ts)


def reintegrate(api, scene, view, rs, batch, rank=0, world=1, chunk_blocks=64, all_gather=None, timers=None,
                force_collective=False, stored_lists=False, batched=False):
    """Run the batch on this rank; `all_gather(counts)` performs pack -> collective -> unpack for the per-shard dirty
    block counts (None when world == 1; force_collective runs it even for a single rank, as a plumbing check).
    stored_lists: the frames are ("store", frame_store, slot) keyframes whose fusion-time visible lists sit in the
    store; de-integration then skips its allocation pass (dslam_deprocess_frame_stored) -- the part of the batch that
    every rank would otherwise repeat -- and the list of the re-fusion replaces the stored one.
    batched (implies stored lists, all keyframes in ONE store): the whole loop as one dslam_reintegrate_batch call -- the HIP
    engine then runs it block-major (allocation passes first, every touched voxel block loaded once); same bytes."""
    t0 = time.perf_counter()
    collective = (world > 1 or force_collective) and all_gather is not None
    if collective:
        api.track_dirty(scene, True)
    if world > 1:
        api.set_shard(scene, rank, world, chunk_blocks)
    if batched:
        store = batch.frames[0][1]
        assert all(f[0] == "store" and f[1] is store for f in batch.frames), "a batched run needs all keyframes in one store"
        api.reintegrate_batch(scene, view, rs, store, [f[2] for f in batch.frames], batch.old_poses, batch.new_poses, batch.intr)
    for k in range(0 if batched else len(batch)):
        _update_view(api, view, batch.frames[k], float(k))
        if stored_lists:
            api.deprocess_frame_stored(scene, view, batch.frames[k][1], batch.frames[k][2], batch.old_poses[k], batch.intr)
        else:
            api.deprocess_frame(scene, view, rs, batch.old_poses[k], batch.intr)  # DenseSlam.cpp:390-393
        api.process_frame(scene, view, rs, batch.new_poses[k], batch.intr, is_defusion=True)  # DenseSlam.cpp:401-403
        if stored_lists:
            api.frame_store_put_visible_list(batch.frames[k][1], batch.frames[k][2], scene, rs)
    api.stats(scene, rs)  # synchronises
    t1 = time.perf_counter()
    counts = None
    if collective:
        counts = api.shard_dirty_plan(scene, world, chunk_blocks)  # identical on every rank
        all_gather(counts)
        api.track_dirty(scene, False)
    if world > 1:
        api.set_shard(scene, 0, 1, chunk_blocks)
    t2 = time.perf_counter()
    if timers is not None:
        timers.update(reintegrate_s=t1 - t0, all_gather_s=t2 - t1, total_s=t2 - t0,
                      dirty_blocks=0 if counts is None else int(sum(counts)),
                      gathered_bytes=0 if counts is None else world * max(counts) * BLOCK_BYTES)
    return counts


def make_torch_all_gather(api, scene, dist, engine_sync):
    """pack -> all_gather_into_tensor -> unpack on torch CUDA tensors (HIP engine; backend nccl = RCCL), ordered by the
    ENGINE'S stream alone: torch is told to treat that stream as its current one for the collective (ExternalStream), so the
    collective waits for the pack kernel and the unpack kernel for the collective through stream order -- as the native
    program does (itmlib/tests/reintegrate_rccl.cpp: ncclAllGather on dslam_engine_stream).  Up to round 3 the two sides
    were drained on the host three times per exchange.  One wait at the end (the caller reads timers / the map next)."""
    import torch

    ext = None

    def run(counts):
        nonlocal ext
        world, rank = dist.get_world_size(), dist.get_rank()
        cap = max(1, max(counts))
        dev = torch.device("cuda", torch.cuda.current_device())
        if ext is None:
            ext = torch.cuda.ExternalStream(api.stream(), device=dev)
        with torch.cuda.stream(ext):   # (allocations and the collective are stream-ordered on the engine's stream)
            send = torch.empty((cap, BLOCK_BYTES), dtype=torch.uint8, device=dev)
            recv = torch.empty((world, cap, BLOCK_BYTES), dtype=torch.uint8, device=dev)
            api.shard_dirty_pack(scene, rank, send.data_ptr(), cap)
            dist.all_gather_into_tensor(recv.view(-1), send.view(-1))
            api.shard_dirty_unpack(scene, rank, recv.data_ptr(), cap)
        engine_sync()
        del send, recv
    return run


def make_staged_all_gather(api, scene, dist, engine_sync):
    """HIP engine, collective over HOST memory (backend gloo): pack on the device, stage through the host, unpack on the
    device.  Not a production path -- it is how several ranks that SHARE one GPU (the one-GPU box: RCCL refuses two ranks on
    one device) run the sharded kernels and the exchange across real process boundaries (tests/test_gpu_multirank.py)."""
    import torch

    def run(counts):
        world, rank = dist.get_world_size(), dist.get_rank()
        cap = max(1, max(counts))
        dev = torch.device("cuda", torch.cuda.current_device())
        send = torch.zeros((cap, BLOCK_BYTES), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        api.shard_dirty_pack(scene, rank, send.data_ptr(), cap)
        engine_sync()
        recv_h = torch.empty((world, cap, BLOCK_BYTES), dtype=torch.uint8)
        dist.all_gather_into_tensor(recv_h.view(-1), send.cpu().view(-1))
        recv = recv_h.to(dev)
        torch.cuda.synchronize()
        api.shard_dirty_unpack(scene, rank, recv.data_ptr(), cap)
        engine_sync()
    return run


def make_numpy_all_gather(api, scene, dist):
    """The same on host buffers, for engines whose voxel blocks live in host memory (the CPU oracle under gloo)."""
    import torch

    def run(counts):
        world, rank = dist.get_world_size(), dist.get_rank()
        cap = max(1, max(counts))
        send = np.zeros((cap, BLOCK_BYTES), dtype=np.uint8)
        api.shard_dirty_pack(scene, rank, send.ctypes.data, cap)
        recv = torch.empty((world, cap, BLOCK_BYTES), dtype=torch.uint8)
        dist.all_gather_into_tensor(recv.view(-1), torch.from_numpy(send).view(-1))
        out = np.ascontiguousarray(recv.numpy())
        api.shard_dirty_unpack(scene, rank, out.ctypes.data, cap)
    return run


# ---------------------------------------------------------------------------------------------------------------------
# DenseSlam::OnlineCorrection's scheduler (reference DenseSlam.cpp:298-432) over a device-resident keyframe store.
# The C++ counterpart a maintainer links against is itmlib/DenseSLAM/OnlineCorrection.h; this is the same logic for
# the Python harnesses (bench --reint, the gloo test), engine-agnostic like the rest of this module.
# ---------------------------------------------------------------------------------------------------------------------
def se3_log(T):
    """(translation part u, rotation vector w) of the se(3) logarithm of a 4x4 rigid transform (float64)."""
    T = np.asarray(T, np.float64)
    R, t = T[:3, :3], T[:3, 3]
    c = min(1.0, max(-1.0, (np.trace(R) - 1.0) * 0.5))
    angle = np.arccos(c)
    w = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.linalg.norm(w)
    if s > 1e-12:
        w = w * (angle / s)
    u = t.copy()
    if angle > 1e-9:
        K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        half_cot = 0.5 * angle * np.sin(angle) / (1.0 - np.cos(angle))
        u = t - 0.5 * (K @ t) + ((1.0 - half_cot) / angle ** 2) * (K @ K @ t)
    return u, w


def pose_error(pre_Twc, cur_Twc):
    """sqrt(trace(E W E^T)), W = diag(.5,.5,.5,1), E = hat(log(pre^-1 cur)) (DenseSlam.cpp:322-357)
    = sqrt(|rotation vector|^2 + |translation part|^2)."""
    u, w = se3_log(np.linalg.inv(np.asarray(pre_Twc, np.float64)) @ np.asarray(cur_Twc, np.float64))
    return float(np.sqrt(w @ w + u @ u))


class FusionFrameDatabase:
    """mfusionFrameDataBase (DenseSlam.h:431-433) with the images in a keyframe store on the engine's device."""

    def __init__(self, api, width, height, capacity, pose_to_M=None):
        self.api = api
        self.store = api.create_frame_store(width, height, capacity)
        self.entries = {}  # timestamp -> [Twc, slot, flag]
        self.free = list(range(capacity - 1, -1, -1))
        self.pose_to_M = pose_to_M or (lambda Twc: np.linalg.inv(np.asarray(Twc, np.float64)).astype(np.float32))

    def __len__(self):
        return len(self.entries)

    def _slot_for(self, ts):
        if ts in self.entries:
            return self.entries[ts][1]
        if not self.free:
            raise RuntimeError("keyframe store full")
        return self.free.pop()

    def insert_from_view(self, ts, Twc, view):
        slot = self._slot_for(ts)
        self.api.frame_store_put_view(self.store, slot, view)
        self.entries[ts] = [np.array(Twc, np.float32), slot, 0]
        return slot

    def insert(self, ts, Twc, rgba, depth_mm):
        slot = self._slot_for(ts)
        self.api.frame_store_put(self.store, slot, rgba, depth_mm)
        self.entries[ts] = [np.array(Twc, np.float32), slot, 0]
        return slot

    def _erase(self, ts):
        self.free.append(self.entries.pop(ts)[1])

    def slide_window_pose(self, max_age):
        """DenseSlam::SlideWindowPose (DenseSlam.cpp:284-296): drop the oldest entries until max_age remain."""
        for ts in sorted(self.entries)[:max(0, len(self.entries) - max_age)]:
            self._erase(ts)

    def plan(self, keyframes, correction_num, start_to_correction_num, identity_eps=0.0):
        """The selection half of OnlineCorrection: returns ([(ts, new Twc)] in re-fusion order, [culled ts]).
        keyframes: iterable of (timestamp, Twc, is_bad).  Marks flags like the reference does."""
        errors = {}
        for ts, Twc, bad in keyframes:
            if bad or ts not in self.entries:
                continue
            ent = self.entries[ts]
            ent[2] = 1
            err = pose_error(ent[0], Twc)
            if err <= identity_eps:
                continue  # is_identity_matrix(poseDiff)
            errors[np.float32(err)] = (ts, np.array(Twc, np.float32))  # equal keys overwrite (std::map)
        order = []
        if len(errors) > start_to_correction_num - 1:
            for err in sorted(errors, reverse=True):
                if errors[err][0] in self.entries:
                    order.append(errors[err])
                if len(order) > correction_num - 1:
                    break
        culled = [ts for ts in sorted(self.entries) if self.entries[ts][2] == 0]
        return order, culled

    def enable_visible_lists(self, scene):
        """(extension) room for one visible list per keyframe: what the batched correction de-integrates from"""
        self.api.frame_store_enable_lists(self.store, scene)

    def keep_visible_list(self, ts, scene, rs):
        """after every fusion of keyframe `ts` (itmlib: FusionFrameDataBase / ITMDenseMapper::KeepVisibleList)"""
        self.api.frame_store_put_visible_list(self.store, self.entries[ts][1], scene, rs)

    def online_correction(self, scene, view, rs, intr, keyframes, correction_num, start_to_correction_num,
                          identity_eps=0.0, batched=False):
        """DenseSlam::OnlineCorrection: de-integrate each selected keyframe at its old pose, re-integrate it at the
        optimised one with isDefusion, then take culled keyframes out of the map.  No image leaves the device.
        batched: the re-fusions as ONE dslam_reintegrate_batch call (itmlib: OnlineCorrectionBatched); the keyframes'
        visible lists must have been kept (enable_visible_lists + keep_visible_list)."""
        api = self.api
        order, culled = self.plan(keyframes, correction_num, start_to_correction_num, identity_eps)
        if batched and order:
            ents = [self.entries[ts] for ts, _ in order]
            api.reintegrate_batch(scene, view, rs, self.store, [e[1] for e in ents], [self.pose_to_M(e[0]) for e in ents],
                                  [self.pose_to_M(T) for _, T in order], intr)
            for e, (_, T) in zip(ents, order):
                e[0] = T
            order_loop = []
        else:
            order_loop = order
        for ts, new_Twc in order_loop:
            ent = self.entries[ts]
            api.view_update_from_store(view, self.store, ent[1], timestamp=float(ts))
            api.deprocess_frame(scene, view, rs, self.pose_to_M(ent[0]), intr)
            ent[0] = new_Twc
            api.process_frame(scene, view, rs, self.pose_to_M(ent[0]), intr, is_defusion=True)
        for ts in culled:
            ent = self.entries[ts]
            api.view_update_from_store(view, self.store, ent[1], timestamp=0.0)
            api.deprocess_frame(scene, view, rs, self.pose_to_M(ent[0]), intr)
            self._erase(ts)
        return [ts for ts, _ in order], culled
