"""Sharded global re-integration after a loop closure (BASELINE config 4).

DenseSlam::OnlineCorrection (reference DenseSlam.cpp:298-432) de-integrates every corrected keyframe at its old pose
and re-integrates it at the new one.  A voxel's new value depends only on its own old value and the frame sequence,
so voxel blocks are independent units and the batch shards by voxel-block slot:

  1. every rank holds the full map replica and the same list of (frame, old pose, new pose);
  2. every rank runs the allocation passes of all frames -- deterministic and bit-identical, so hash tables and
     free lists stay identical on all ranks without any exchange;
  3. rank g de-integrates / re-integrates only blocks whose slot chunk (slot // chunk_blocks) % world == g
     (dslam_scene_set_shard; round-robin chunks balance the load because slots are handed out top-down), in keyframe
     order -- so every voxel sees exactly the update sequence of the single-GPU run;
  4. ONE collective at the end: all-gather of the used slot range [lo, N) of the voxel-block array.  Each rank packs
     its chunks (a strided view) into a contiguous send buffer, all_gather_into_tensor (RCCL over xGMI on GPUs:
     direct peer-to-peer all-gather keeps all links busy, time ~ shard_bytes / link rate), and the result is
     scattered back into the chunk-interleaved layout.

The result is byte-identical to the unsharded run (tests/test_multigpu_gloo.py).  The module is engine-agnostic:
`api` is a bound C ABI (`CApi`), either the HIP library (voxel blocks live in a torch CUDA tensor handed to
dslam_scene_create as the external voxel-block buffer; collective = NCCL/RCCL) or -- for the world_size-2 CPU tests
-- the oracle (voxel blocks copied through numpy; collective = gloo).
"""
import time

import numpy as np

BLOCK_BYTES = 512 * 8


def plan_region(last_free_block_id, num_local_blocks, world, chunk_blocks):
    """Smallest slot range [lo, N) that covers the used slots and is a whole number of chunk groups
    (chunk_blocks * world slots).  Requires N to be a multiple of chunk_blocks * world."""
    group = chunk_blocks * world
    if num_local_blocks % group:
        raise ValueError("num_local_blocks must be a multiple of chunk_blocks * world")
    used = num_local_blocks - 1 - last_free_block_id
    groups = -(-used // group)
    return num_local_blocks - groups * group, groups


class Batch:
    """The corrected keyframes: frames (("dev", rgba_ptr, depth_ptr) or ("host", rgba, depth_mm)) plus old / new
    world->camera poses and the intrinsics."""

    def __init__(self, frames, old_poses, new_poses, intr):
        self.frames, self.old_poses, self.new_poses, self.intr = frames, old_poses, new_poses, intr

    def __len__(self):
        return len(self.old_poses)


def _update_view(api, view, frame, ts):
    if frame[0] == "dev":
        api.view_update_device(view, frame[1], frame[2], timestamp=ts)
    else:
        api.view_update(view, frame[1], frame[2], timestamp=ts)


def reintegrate(api, scene, view, rs, batch, rank=0, world=1, chunk_blocks=64, all_gather=None, timers=None,
                force_collective=False):
    """Run the batch on this rank; `all_gather(lo, groups)` performs the collective (None when world == 1;
    force_collective runs it even for a single rank, as a plumbing check)."""
    t0 = time.perf_counter()
    collective = (world > 1 or force_collective) and all_gather is not None
    if world > 1:
        api.set_shard(scene, rank, world, chunk_blocks)
    for k in range(len(batch)):
        _update_view(api, view, batch.frames[k], float(k))
        api.deprocess_frame(scene, view, rs, batch.old_poses[k], batch.intr)  # DenseSlam.cpp:390-393
        api.process_frame(scene, view, rs, batch.new_poses[k], batch.intr, is_defusion=True)  # DenseSlam.cpp:401-403
    st = api.stats(scene, rs)  # synchronises; identical on every rank
    t1 = time.perf_counter()
    lo = groups = None
    if collective:
        lo, groups = plan_region(st["last_free_block_id"], scene.params.num_local_blocks, world, chunk_blocks)
        all_gather(lo, groups)
    if world > 1:
        api.set_shard(scene, 0, 1, chunk_blocks)
    t2 = time.perf_counter()
    if timers is not None:
        timers.update(reintegrate_s=t1 - t0, all_gather_s=t2 - t1, total_s=t2 - t0,
                      gathered_bytes=0 if not collective else groups * world * chunk_blocks * BLOCK_BYTES)
    return lo, groups


def make_torch_all_gather(voxel_tensor, dist, chunk_blocks, engine_sync):
    """Collective over a torch uint8 CUDA tensor that IS the scene's voxel-block array (HIP engine)."""
    import torch

    def run(lo, groups):
        world, rank = dist.get_world_size(), dist.get_rank()
        chunk_bytes = chunk_blocks * BLOCK_BYTES
        region = voxel_tensor[lo * BLOCK_BYTES:].view(groups, world, chunk_bytes)
        engine_sync()  # the engine's kernels run on its own stream
        send = region[:, rank, :].contiguous()
        recv = torch.empty((world, groups, chunk_bytes), dtype=torch.uint8, device=voxel_tensor.device)
        dist.all_gather_into_tensor(recv.view(-1), send.view(-1))
        region.copy_(recv.permute(1, 0, 2))
        torch.cuda.synchronize()
    return run


def make_numpy_all_gather(api, scene, dist, chunk_blocks):
    """Collective for engines whose voxel blocks are not a torch tensor (the CPU oracle under gloo)."""
    import torch

    def run(lo, groups):
        world, rank = dist.get_world_size(), dist.get_rank()
        n = groups * world * chunk_blocks
        vox = api.download_voxel_blocks(scene, lo, n)
        region = vox.view(np.uint8).reshape(groups, world, chunk_blocks * BLOCK_BYTES)
        send = torch.from_numpy(np.ascontiguousarray(region[:, rank, :]))
        recv = torch.empty((world,) + tuple(send.shape), dtype=torch.uint8)
        dist.all_gather_into_tensor(recv.view(-1), send.view(-1))
        merged = np.ascontiguousarray(recv.numpy().transpose(1, 0, 2))
        api.upload_voxel_blocks(scene, lo, merged.view(vox.dtype).reshape(n, 512))
    return run
