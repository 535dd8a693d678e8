#!/usr/bin/env python3
"""Experiment: do two renderers on one GPU (two streams, half of the samples each) beat one?  Kernel-overlap potential:
the traversal launches are VALU bound, kShade / kRaygen / kAccumulateRuns are memory bound.  `extra_lds` lowers the residency of
the persistent traversal kernels (4096: 5 workgroups per CU, 9216: 4) so that the other stream's workgroups find registers.

usage: gpu_concurrent.py [total spp = 320] [rounds = 3]
"""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import rayfinder_amd as rf
from rayfinder_amd import scenes
total = int(sys.argv[1]) if len(sys.argv) > 1 else 320
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pt, info = scenes.atrium()
W, H, b = 1920, 1080, 8
cam = rf.fly_camera(W, H)
def params(spp, expo):
    return rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), expo)
def make(spp):
    return rf.ReferencePathTracer(params(spp, 0.25), pt.scene())
def run(rs, spp):
    for r in rs: r.reset_stats()
    t0 = time.perf_counter()
    th = [threading.Thread(target=lambda r=r: (r.render(spp), r.synchronize())) for r in rs]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    rays = sum(r.stats()["closest_rays"] + r.stats()["shadow_rays"] for r in rs)
    return rays / dt * 1e-6
expo = 0.5
for n, lds in ((1, 0), (1, 4096), (2, 0), (2, 4096), (2, 9216), (3, 4096)):
    spp = total // n
    rs = [make(spp) for _ in range(n)]
    for r in rs:
        if lds: r.set_option("extra_lds", lds)
        r.render(spp); r.synchronize()
    best = []
    for _ in range(rounds):
        expo *= 0.99
        for r in rs: r.set_render_parameters(params(spp, expo))  # restarts the accumulation
        best.append(run(rs, spp))
    print(f"{n} renderer(s) x {spp} spp, extra_lds={lds}: median {np.median(best):.1f} best {max(best):.1f} Mrays/s", flush=True)
    for r in rs: r.close()
