#!/usr/bin/env python3
"""Throughput vs paths-in-flight per batch (atrium 1080p, 8 bounces)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import rayfinder_amd as rf
from rayfinder_amd import scenes
pt, info = scenes.atrium()
W, H, b, spp = 1920, 1080, 8, 1024
cam = rf.fly_camera(W, H)
for mpaths in [512, 1024, 512]:
    r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), 0.25), pt.scene(), max_paths_in_flight=mpaths << 20)
    r.render(spp); r.synchronize()
    best = 0
    for rep in range(2):
        r.set_render_parameters(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), 0.5 + rep))
        r.reset_stats()
        t0 = time.perf_counter(); r.render(spp); r.synchronize(); dt = time.perf_counter() - t0
        s = r.stats(); best = max(best, (s["closest_rays"] + s["shadow_rays"]) / dt * 1e-6)
    print(f"paths in flight {mpaths:4d} Mi: {best:8.1f} Mrays/s")
    r.close()
