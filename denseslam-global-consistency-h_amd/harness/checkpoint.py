"""Map checkpoint / restore (SURVEY.md section 5 "Checkpoint / resume": the reference has none for the map; next-row
item "raw dump of hash table + VBA").  Engine-agnostic like the rest of the harness: works through the C ABI's bulk
read-back and upload entry points, so a map saved from the HIP engine restores into the HIP engine or the CPU oracle
and vice versa.

The file is a numpy .npz: scene parameters, the hash table, both free lists with their stack tops, and the voxel
blocks of the slot range that is in use (slots are handed out from the top of the pool, so the used range is
[min used slot, N)).  Not saved: the render state (rebuilt by the next AllocateSceneFromDepth), the visible-list
history rings of decay / sliding window (a restored map starts with an empty window), and blocks parked in the host
swap store (flush with save_to_global_memory + disable swapping first, or keep them out of the checkpoint).
"""
import numpy as np

PARAM_FIELDS = ("voxel_size", "mu", "max_w", "frustum_min", "frustum_max", "stop_integrating_at_max_w", "num_local_blocks",
                "num_buckets", "num_excess", "use_swapping", "history_words")


def save_map(api, scene, path):
    st = api.stats(scene)
    table = api.download_hash_table(scene)
    n = scene.params.num_local_blocks
    used = table["ptr"][table["ptr"] >= 0]
    first = int(used.min()) if used.size else n
    blocks = api.download_voxel_blocks(scene, first, n - first) if first < n else np.zeros((0, 512), dtype=table.dtype)
    np.savez_compressed(
        path, params=np.array([getattr(scene.params, f) for f in PARAM_FIELDS], dtype=np.float64), hash=table.view(np.uint8),
        alloc_list=api.download_allocation_list(scene), excess_list=api.download_excess_list(scene),
        tops=np.array([st["last_free_block_id"], st["last_free_excess_id"]], dtype=np.int64),
        first_block=np.array([first], dtype=np.int64), blocks=np.ascontiguousarray(blocks).view(np.uint8))
    return {"used_blocks": int(used.size), "first_block": first}


def load_map(api, pkg, path):
    """Create a scene on `api`'s engine and fill it from the checkpoint; returns the scene."""
    with np.load(path) as z:
        vals = z["params"]
        kw = {f: (float(v) if f in ("voxel_size", "mu", "frustum_min", "frustum_max") else int(v)) for f, v in zip(PARAM_FIELDS, vals)}
        scene = api.create_scene(pkg.SceneParams(**kw))
        table = z["hash"].view(pkg.HASH_ENTRY_DTYPE)
        api.upload_scene_state(scene, hash_table=table, allocation_list=z["alloc_list"], last_free_block_id=int(z["tops"][0]),
                               excess_list=z["excess_list"], last_free_excess_id=int(z["tops"][1]))
        first = int(z["first_block"][0])
        blocks = z["blocks"].view(pkg.VOXEL_DTYPE).reshape(-1, 512)
        if blocks.shape[0]:
            api.upload_voxel_blocks(scene, first, blocks)
    return scene
