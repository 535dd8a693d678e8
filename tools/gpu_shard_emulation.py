#!/usr/bin/env python3
"""Strong-scaling emulation on ONE GPU: the tile shard of rank r of a world of N (what one rank of the N-GPU bench traces),
timed like bench.py's timed region (320 spp, one exchange-free frame).  Projected speed-up of N GPUs = N * rays/s of a 1/N
shard / rays/s of the whole frame -- everything except the RCCL exchange itself (33 MB at frame end)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import rayfinder_amd as rf
from rayfinder_amd import scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 320
W, H, B = 1920, 1080, 8
pt, info = scenes.atrium()
cam = rf.fly_camera(W, H); sky = rf.make_sky()
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
    print(f"world {world}: per-rank time (ranks {[x[0] for x in rates]}) {[round(x[2]*1e3,1) for x in rates]} ms, rays/rank {[int(x[1]/1e6) for x in rates]} M; "
          f"projected speed-up {full / worst:5.2f}x (efficiency {full / worst / world:5.3f})")
