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
    # the exchange as the C++ posts it: rf_gather_plan's sends / receives, executed over gloo point-to-point
    from rayfinder_amd.sharding import exchange_with_plan
    root = world - 1
    via_plan = exchange_with_plan(torch.from_numpy(compact), w, h, rank, world, root)
    if rank == root:
        np.save(os.path.join(outdir, f"plan_{world}.npy"), via_plan)
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
    via_plan = np.load(tmp_path / f"plan_{world}.npy")          # root = the LAST rank: the plan is not special-cased for rank 0
    assert np.array_equal(via_plan.view(np.uint32), want.view(np.uint32))


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


@pytest.mark.parametrize("w,h,world", [(1920, 1080, 2), (1920, 1080, 4), (1920, 1080, 8), (3840, 2160, 8), (100, 70, 3), (33, 1, 2), (64, 64, 7)])
def test_gather_plan_of_every_rank_matches_the_staging_layout(w, h, world):
    """The list of (peer, offset, count) operations the C++ exchange posts (rf_gather_plan == TileComm::gatherFrame's loop),
    for EVERY rank of the world and every root: each send has exactly one matching receive of the same size, the root's
    receives land on disjoint staging ranges that together with its own shard tile the staging area, and a dry run of the
    plan over host buffers reproduces the frame."""
    import rayfinder_amd as rf
    first, slot, owner = rf.gather_layout(w, h, world)
    n_tiles = ((w + 31) // 32) * ((h + 31) // 32)
    tiles_of = [rf.tiles_for_rank(w, h, r, world) for r in range(world)]
    for root in sorted({0, world - 1, world // 2}):
        plans = [rf.gather_plan(w, h, world, r, root) for r in range(world)]
        sends = {(r, int(p)): int(c) for r in range(world) for s_, p, o, c in plans[r].tolist() if s_}
        recvs = {(int(p), root): (int(o), int(c)) for s_, p, o, c in plans[root].tolist() if not s_}
        assert set(sends) == set(recvs), "a send without a receive (or the reverse) would hang the group"
        for key, cnt in sends.items():
            assert recvs[key][1] == cnt == len(tiles_of[key[0]])
            assert key[0] != root and key[1] == root
        for r in range(world):
            if r != root:
                assert all(s_ == 1 for s_, *_ in plans[r].tolist()) and len(plans[r]) == (1 if len(tiles_of[r]) else 0)
                assert all(o == 0 for _, _, o, _ in plans[r].tolist())           # a rank sends its compact buffer from the start
        covered = np.zeros(n_tiles, np.int32)
        for (p, _), (off, cnt) in recvs.items():
            assert off == first[p]
            covered[off:off + cnt] += 1
        covered[first[root]:first[root + 1]] += 1                              # read in place by the un-tile kernel
        assert (covered == 1).all()
        # dry run: tile t of the frame carries the value t; staging filled by the plan must map back through tile_slot
        staging = np.full(n_tiles, -1, np.int64)
        for (p, _), (off, cnt) in recvs.items():
            staging[off:off + cnt] = tiles_of[p]
        staging[first[root]:first[root + 1]] = tiles_of[root]
        assert np.array_equal(staging[slot], np.arange(n_tiles))
    # loopback (the world-size-1 self-test of the RCCL path): the root also sends to itself
    lp = rf.gather_plan(w, h, world, 0, 0, loopback=True)
    assert [tuple(x) for x in lp.tolist() if x[0]] == [(1, 0, 0, len(tiles_of[0]))]
    assert (0, 0, int(first[0]), len(tiles_of[0])) in [tuple(x) for x in lp.tolist()]


def _gpu_worker(rank, world, port, w, h, outdir):
    """Two processes share the one GPU of the box (no RCCL between them: gloo carries the plan's sends / receives);
    each renders ITS tile shard with the product and the root un-tiles what the C++ plan says goes where."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rayfinder_amd as rf
    from rayfinder_amd.sharding import exchange_with_plan, tile_major
    pt = rf.PtFormat.from_gltf(os.path.join(ROOT, "tests", "golden", "Duck.glb"))
    spp, bounces = 4, 3
    params = rf.make_render_parameters(w, h, rf.fly_camera(w, h), spp, bounces, rf.make_sky(), 0.25)
    r = rf.ReferencePathTracer(params, pt.scene())
    r.set_tile_shard(rank, world)
    r.render(spp)
    img, acc = r.read_accumulation()
    assert acc == spp
    tiles = r.shard_tiles()
    assert np.array_equal(tiles, rf.tiles_for_rank(w, h, rank, world))
    image = exchange_with_plan(torch.from_numpy(tile_major(img, tiles, w, h)), w, h, rank, world, 0)
    if rank == 0:
        np.save(os.path.join(outdir, "product_gathered.npy"), image)
        r.set_tile_shard(0, 1)
        r.render(spp)
        np.save(os.path.join(outdir, "product_whole.npy"), r.read_accumulation()[0])
    r.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_gloo_plan_exchange_of_product_rendered_shards(tmp_path):
    w, h, world = 200, 136, 2
    mp.spawn(_gpu_worker, args=(world, _free_port(), w, h, str(tmp_path)), nprocs=world, join=True)
    got, want = np.load(tmp_path / "product_gathered.npy"), np.load(tmp_path / "product_whole.npy")
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


# ------------------------------------------------------------------ round 4: bench.py --gpus N is a launcher by itself
def _run_bench(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


def _no_gpu_here():
    return not torch.cuda.is_available()


@pytest.mark.skipif(not _no_gpu_here(), reason="checks the no-GPU exit of every self-launched rank")
def test_bench_gpus_2_without_a_launcher_starts_two_ranks_itself():
    """`python bench.py --gpus 2` (the shape of the driver's command, no WORLD_SIZE around it) must not run one rank and print
    n_gpus: 1: it re-launches itself as two ranks under torch.distributed.run --standalone on 127.0.0.1.  In this container
    both ranks get as far as the device check and exit non-zero, each naming itself; no JSON line is printed."""
    rc, out, err = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert rc != 0
    assert "launching 2 ranks" in err and "torch.distributed.run" in err and "--nproc-per-node=2" in err
    assert "rank 0 of 2: needs 2 MI355X GPU(s)" in err and "rank 1 of 2: needs 2 MI355X GPU(s)" in err
    assert "n_gpus" not in out


def test_bench_refuses_a_world_that_disagrees_with_gpus():
    """Under a launcher, WORLD_SIZE ranks exist whatever --gpus says: a line claiming another n_gpus would be a wrong scaling
    point, so the rank exits non-zero and says which of the two to change."""
    rc, out, err = _run_bench(["--gpus", "4"], env_extra=dict(WORLD_SIZE="2", RANK="1", LOCAL_RANK="1"), drop=())
    assert rc != 0 and "--gpus 4 but the launcher started WORLD_SIZE=2" in err and out.strip() == ""


@pytest.mark.skipif(not _no_gpu_here(), reason="checks the no-GPU exit")
def test_bench_single_rank_exits_loudly_without_a_gpu():
    rc, out, err = _run_bench(["--gpus", "1"])
    assert rc != 0 and "no GPU visible" in err and "launching" not in err and out.strip() == ""
