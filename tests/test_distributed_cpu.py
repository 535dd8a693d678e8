"""world_size-2 (and 3) gloo tests of the multi-GPU path's host logic, on CPU: every rank takes
its tile shard, produces its compact tile-major buffer, rank 0 gathers (the same
torch.distributed.gather call bench.py issues over RCCL) and un-tiles; the result must equal the
single-rank image bit for bit.  Pixel values come from the oracle (allowed in tests), so this
also checks that sharding cannot change any pixel: each pixel's estimate depends only on
(x, y, sample index, scene, params)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rayfinder_amd as rf
    from rayfinder_amd.sharding import assemble, assemble_with_layout, gather_device, shard_layout
    from conftest import DuckOracle
    from oracle import orc

    d = DuckOracle()
    spp, bounces = 2, 2
    rp = orc.make_render_params(w, h, orc.default_pt_camera(w, h), spp, bounces, 0.25, orc.aligned_sky_state())
    tiles, max_tiles = shard_layout(w, h, rank, world)
    tx = (w + 31) // 32
    compact = np.zeros((max_tiles * 1024, 4), np.float32)
    for t, tid in enumerate(tiles):
        x0, y0 = (tid % tx) * 32, (tid // tx) * 32
        img, _ = orc.render(d.scene, rp, 0, spp, x0, y0, min(x0 + 32, w), min(y0 + 32, h))
        for k in range(1024):
            block, lane = k >> 6, k & 63
            x = x0 + (block & 3) * 8 + (lane & 7); y = y0 + (block >> 2) * 8 + (lane >> 3)
            if x < w and y < h:
                compact[t * 1024 + k] = img[y, x]
    parts = gather_device(torch.from_numpy(compact), rank, world)
    if rank == 0:
        image = assemble(parts, w, h, world)
        np.save(os.path.join(outdir, f"gathered_{world}.npy"), image)
        # the staging layout the C++ RCCL exchange uses (rf_gather_layout / kUntile's mapping)
        np.save(os.path.join(outdir, f"layout_{world}.npy"), assemble_with_layout(parts, w, h, world))
        if world == 1 or not os.path.exists(os.path.join(outdir, "whole.npy")):
            whole, _ = orc.render(d.scene, rp, 0, spp)
            np.save(os.path.join(outdir, "whole.npy"), whole)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_tile_gather_reassembles_the_image(world, tmp_path):
    w, h = 100, 70      # 4 x 3 tiles, ragged right/bottom edge
    mp.spawn(_worker, args=(world, _free_port(), w, h, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / f"gathered_{world}.npy")
    want = np.load(tmp_path / "whole.npy")
    assert got.shape == (h, w, 4)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    via_layout = np.load(tmp_path / f"layout_{world}.npy")
    assert np.array_equal(via_layout.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("w,h,world", [(1920, 1080, 8), (100, 70, 3), (64, 64, 5), (33, 1, 2)])
def test_gather_layout_is_a_partition_of_the_staging_area(w, h, world):
    import rayfinder_amd as rf
    first, slot, owner = rf.gather_layout(w, h, world)
    n = ((w + 31) // 32) * ((h + 31) // 32)
    assert first[0] == 0 and first[-1] == n and (np.diff(first.astype(np.int64)) >= 0).all()
    assert np.array_equal(np.sort(slot), np.arange(n))                     # every staging tile used exactly once
    for r in range(world):
        tiles = rf.tiles_for_rank(w, h, r, world)
        assert (owner[tiles] == r).all()
        assert np.array_equal(slot[tiles], first[r] + np.arange(len(tiles)))   # a rank's tiles in ascending id = its send order
