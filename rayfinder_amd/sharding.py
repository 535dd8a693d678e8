"""Multi-GPU test harness: the tile-shard arithmetic of the frame-end exchange, driven through torch.distributed.

The product's exchange is C++ (rf_renderer_gather_frame: RCCL ncclSend/ncclRecv + a device un-tile, rf_comm.hip);
this module restates its host-visible half -- which tile goes where -- so that the world_size > 1 path can run
on CPU under gloo (tests/test_distributed_cpu.py).  The path shards by image tiles (every pixel's estimate depends
only on (x, y, sample index, scene, params): reference_path_tracer.wgsl:42-57), the scene is replicated per GPU,
and the only exchange is the gather of the compact per-rank accumulation buffers to rank 0.  No reduction, no ring.
"""
import numpy as np

from . import gather_layout, gather_plan, tiles_for_rank, untile


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


def assemble_with_layout(parts, width, height, world_size):
    """Rank 0: what rf_comm.hip does on the device -- shards stored rank after rank in a staging area
    (rf_gather_layout), then one pass per tile of the frame: staging[tile_slot[tile]] -> row-major pixels."""
    first, slot, owner = gather_layout(width, height, world_size)
    staging = np.zeros((int(first[-1]) * 1024, 4), np.float32)
    for r, part in enumerate(parts):
        n = int(first[r + 1] - first[r])
        staging[int(first[r]) * 1024:int(first[r + 1]) * 1024] = part.detach().cpu().numpy()[: n * 1024]
    tiles_x = (width + 31) // 32
    image = np.zeros((height, width, 4), np.float32)
    w = np.arange(1024)
    block, lane = w >> 6, w & 63
    dx, dy = (block & 3) * 8 + (lane & 7), (block >> 2) * 8 + (lane >> 3)
    for tile in range(len(slot)):
        assert first[owner[tile]] <= slot[tile] < first[owner[tile] + 1]
        x, y = (tile % tiles_x) * 32 + dx, (tile // tiles_x) * 32 + dy
        ok = (x < width) & (y < height)
        image[y[ok], x[ok]] = staging[int(slot[tile]) * 1024 + w[ok]]
    return image


def exchange_with_plan(compact, width, height, rank, world_size, root=0, group=None):
    """The frame-end exchange exactly as the C++ posts it (rf_gather_plan = the list TileComm::gatherFrame executes:
    one receive per peer into the root's staging area, one send per non-root rank), driven through torch.distributed
    point-to-point calls so that it runs on CPU under gloo.  Root: the row-major (H, W, 4) image; others: None."""
    import torch
    import torch.distributed as dist

    first, slot, owner = gather_layout(width, height, world_size)
    plan = gather_plan(width, height, world_size, rank, root)
    staging = torch.zeros((int(first[-1]) * 1024, 4), dtype=torch.float32) if rank == root else None
    reqs = []
    for is_send, peer, off, cnt in plan.tolist():
        if is_send:
            reqs.append(dist.isend(compact[off * 1024:(off + cnt) * 1024].contiguous(), dst=peer, group=group))
        else:
            reqs.append(dist.irecv(staging[off * 1024:(off + cnt) * 1024], src=peer, group=group))
    for r in reqs:
        r.wait()
    if rank != root:
        return None
    # the root's own shard is read in place (no loopback)
    n_own = int(first[root + 1] - first[root])
    staging[int(first[root]) * 1024:int(first[root + 1]) * 1024] = compact[: n_own * 1024]
    tiles_x = (width + 31) // 32
    image = np.zeros((height, width, 4), np.float32)
    st = staging.numpy()
    w = np.arange(1024)
    block, lane = w >> 6, w & 63
    dx, dy = (block & 3) * 8 + (lane & 7), (block >> 2) * 8 + (lane >> 3)
    for tile in range(len(slot)):
        x, y = (tile % tiles_x) * 32 + dx, (tile // tiles_x) * 32 + dy
        ok = (x < width) & (y < height)
        image[y[ok], x[ok]] = st[int(slot[tile]) * 1024 + w[ok]]
    return image


def tile_major(image, tiles, width, height, max_tiles=None):
    """Row-major (H, W, 4) image -> the compact tile-major buffer a rank's renderer holds for `tiles`
    (32x32 tiles of 8x8-pixel blocks; pixels outside the frame are zero)."""
    out = np.zeros(((max_tiles or len(tiles)) * 1024, 4), np.float32)
    tiles_x = (width + 31) // 32
    w = np.arange(1024)
    block, lane = w >> 6, w & 63
    dx, dy = (block & 3) * 8 + (lane & 7), (block >> 2) * 8 + (lane >> 3)
    for t, tid in enumerate(tiles):
        x, y = (int(tid) % tiles_x) * 32 + dx, (int(tid) // tiles_x) * 32 + dy
        ok = (x < width) & (y < height)
        out[t * 1024 + w[ok]] = image[y[ok], x[ok]]
    return out
