#!/usr/bin/env python3
"""Experiment: do two renderers on one GPU (two streams) beat one?  Kernel-overlap potential."""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rayfinder_amd as rf
from rayfinder_amd import scenes
pt, info = scenes.atrium()
W, H, b = 1920, 1080, 8
cam = rf.fly_camera(W, H)
def make(spp):
    return rf.ReferencePathTracer(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), 0.25), pt.scene())
def run(rs, spp):
    for r in rs: r.reset_stats()
    t0 = time.perf_counter()
    th = [threading.Thread(target=lambda r=r: (r.render(spp), r.synchronize())) for r in rs]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    rays = sum(r.stats()["closest_rays"] + r.stats()["shadow_rays"] for r in rs)
    return rays / dt * 1e-6
one = make(128); one.render(32); one.synchronize()
one.set_render_parameters(rf.make_render_parameters(W, H, cam, 128, b, rf.make_sky(), 0.5))
print("one renderer, 128 spp:", round(run([one], 128), 1), "Mrays/s")
one.close()
for blocks in (0, 768, 1024):
    two = [make(64), make(64)]
    for r in two:
        if blocks: r.set_option("persistent_blocks", blocks)
        r.render(32); r.synchronize()
        r.set_render_parameters(rf.make_render_parameters(W, H, cam, 64, b, rf.make_sky(), 0.5))
    print(f"two renderers x 64 spp, persistent_blocks={blocks or 'default'}:", round(run(two, 64), 1), "Mrays/s")
    for r in two: r.close()
