#!/usr/bin/env python3
"""Strong-scaling emulation on ONE GPU: the tile shard of rank r of a world of N (what one rank of the N-GPU bench traces),
timed like bench.py's timed region (320 spp).  Projected speed-up of N GPUs = time of the whole frame / (time of the slowest 1/N shard + the
frame-end exchange).  The exchange is PRICED (round 5): the product's own rf_renderer_gather_frame at world size 1 -- the 33 MB of the whole frame
sent to itself through RCCL and un-tiled on the device, HIP events around it (rf_comm_last_exchange_ms).  At world N the root receives (N - 1) / N of
those bytes over up to 7 xGMI links at once and un-tiles the same 33 MB, so the loop-back figure (every byte through one device's copy engines,
twice) is an upper bound for what a rank waits; it does not contain RCCL's bring-up (outside the timed region in bench.py) or link contention
(unmeasurable on one GPU)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import rayfinder_amd as rf
from rayfinder_amd import scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 320
W, H, B = 1920, 1080, 8
pt, info = scenes.atrium()
cam = rf.fly_camera(W, H); sky = rf.make_sky()
# ---- the exchange at world 1 (loop-back), median of 5
exchange_ms = None
try:
    comm = rf.TileComm(rf.comm_unique_id(), 0, 1, 0)
    r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, cam, 16, B, sky, 0.25), pt.scene())
    r.render(16); r.synchronize()
    samples = []
    for _ in range(6):
        r.gather_frame(comm, 0, loopback=True); r.synchronize()
        samples.append(comm.last_exchange_ms())
    exchange_ms = sorted(samples[1:])[2]          # (the first one also sets up RCCL's connections)
    print(f"exchange at world 1 (33 MB to itself + device un-tile): {[round(x, 3) for x in samples]} ms -> {exchange_ms:.3f} ms")
    r.close(); comm.close()
except Exception as e:  # noqa: BLE001
    print("exchange not measured:", e)
full = None
for world in (1, 2, 4, 8):
    rates = []
    for rank in sorted({0, world // 2, world - 1}):
        r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, cam, spp, B, sky, 0.25), pt.scene())
        r.set_tile_shard(rank, world)
        r.set_option("reserve_samples", spp)
        r.render(32); r.synchronize()
        r.set_render_parameters(rf.make_render_parameters(W, H, cam, spp, B, sky, 0.5))
        r.set_timing(True); r.reset_stats()
        t0 = time.perf_counter(); r.render(spp); r.synchronize(); dt = time.perf_counter() - t0
        s = r.stats()
        rays = s["closest_rays"] + s["shadow_rays"]
        rates.append((rank, rays, dt))
        r.close()
    worst = max(dt for _, _, dt in rates)
    total_rays = sum(rays for _, rays, _ in rates) / len(rates) * world
    if world == 1: full = worst
    ex = (exchange_ms or 0.0) * 1e-3 if world > 1 else 0.0
    print(f"world {world}: per-rank time (ranks {[x[0] for x in rates]}) {[round(x[2]*1e3,1) for x in rates]} ms, rays/rank {[int(x[1]/1e6) for x in rates]} M; "
          f"projected speed-up {full / worst:5.2f}x without the exchange, {full / (worst + ex):5.2f}x with it ({ex * 1e3:.2f} ms; efficiency {full / (worst + ex) / world:5.3f})")
