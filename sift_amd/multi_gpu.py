"""Image-level data parallelism: one process per GPU, no data-path collective.

The reference handles images independently (Sift/src/sift.cu:68-134 walks the
list two at a time and nothing but the per-slot counters is shared), so a batch
shards by image index.  Image i goes to rank i mod world; every rank extracts
its own images; the per-image point lists are gathered to rank 0 on the host in
image order.  A single image is never split.
"""
import numpy as np


def shard_indices(n_images, rank, world):
    """Indices of the images rank `rank` of `world` works on (i mod world == rank)."""
    if not (0 <= rank < world):
        raise ValueError("rank outside world")
    return list(range(rank, n_images, world))


def extract_sharded(images, width, extract_fn, rank, world):
    """Runs extract_fn(local_images, width) -> (points[n_local, cap], counts[n_local])
    on this rank's shard.  Returns {image index: points[:count]}."""
    idx = shard_indices(len(images), rank, world)
    if not idx:
        return {}
    pts, counts = extract_fn(np.ascontiguousarray(images[idx]), width)
    return {i: np.array(pts[j][: counts[j]]) for j, i in enumerate(idx)}


def gather_in_image_order(local, n_images, dist=None, dst=0):
    """Host-side gather of the per-image results to rank `dst` (None elsewhere).
    `dist` is torch.distributed (any backend) or None for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        parts = [local]
    else:
        parts = [None] * dist.get_world_size() if dist.get_rank() == dst else None
        dist.gather_object(local, parts, dst=dst)
        if dist.get_rank() != dst:
            return None
    merged = {}
    for part in parts:
        for i, v in part.items():
            if i in merged:
                raise RuntimeError(f"image {i} extracted twice")
            merged[i] = v
    missing = [i for i in range(n_images) if i not in merged]
    if missing:
        raise RuntimeError(f"images never extracted: {missing[:8]}")
    return [merged[i] for i in range(n_images)]
