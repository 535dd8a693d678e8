"""Multi-GPU plumbing: tile shards + one gather at frame end.

The path shards by image tiles (every pixel's estimate depends only on (x, y, sample index, scene,
params): reference_path_tracer.wgsl:42-57), the scene is replicated per GPU, and the only exchange
is a gather of the compact per-rank accumulation buffers to rank 0 -- RCCL over xGMI when the
process group backend is "nccl", gloo in the CPU tests.  No reduction, no ring.
"""
import numpy as np

from . import tiles_for_rank, untile


def shard_layout(width, height, rank, world_size):
    """-> (this rank's tile ids, max tiles over ranks). Buffers are padded to max so that the
    gather is a single fixed-size collective."""
    tiles = tiles_for_rank(width, height, rank, world_size)
    n_tiles = ((width + 31) // 32) * ((height + 31) // 32)
    max_tiles = (n_tiles + world_size - 1) // world_size
    return tiles, max_tiles


def gather_device(compact, rank, world_size, group=None):
    """The frame-end exchange: one fixed-size gather of the compact per-rank buffers into rank 0's
    device memory (RCCL over xGMI under the "nccl" backend).  Returns the list of per-rank tensors
    on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist

    if world_size == 1:
        return [compact]
    parts = [torch.empty_like(compact) for _ in range(world_size)] if rank == 0 else None
    dist.gather(compact, parts, dst=0, group=group)
    return parts


def assemble(parts, width, height, world_size):
    """Rank 0: per-rank compact buffers -> row-major (H, W, 4) numpy image (host side)."""
    image = np.zeros((height, width, 4), np.float32)
    for r, part in enumerate(parts):
        tiles = tiles_for_rank(width, height, r, world_size)
        untile(part.detach().cpu().numpy()[: len(tiles) * 1024], tiles, width, height, image)
    return image


def gather_image(compact, width, height, rank, world_size, group=None):
    """gather_device + assemble.  Returns the image on rank 0, None elsewhere."""
    parts = gather_device(compact, rank, world_size, group)
    return assemble(parts, width, height, world_size) if rank == 0 else None
