#!/usr/bin/env python3
"""A/B of renderer options on the atrium: tools/gpu_opt.py name=value[,name=value...] ... (each argument one variant; '-' = defaults)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import rayfinder_amd as rf
from rayfinder_amd import scenes
variants = sys.argv[1:] or ["-"]
spp, rounds = 32, 3
pt, info = scenes.atrium()
W, H, b = 1920, 1080, 8
cam = rf.fly_camera(W, H)
r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), 0.25), pt.scene())
r.render(spp); r.synchronize()
defaults = {}
ref = None; expo = 1.0; res = {v: [] for v in variants}
for rd in range(rounds):
    for v in variants:
        opts = dict(kv.split("=") for kv in v.split(",")) if v != "-" else {}
        for k, val in opts.items(): r.set_option(k, int(val))
        expo *= 0.99
        r.set_render_parameters(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), expo))
        r.set_timing(True); r.reset_stats()
        t0 = time.perf_counter(); r.render(spp); r.synchronize(); dt = time.perf_counter() - t0
        s = r.stats()
        res[v].append(((s["closest_rays"] + s["shadow_rays"]) / dt * 1e-6, s["ms_closest"], s["ms_shadow"], s["ms_shade"]))
        img, _ = r.read_accumulation()
        if ref is None: ref = img
        elif not np.array_equal(img.view(np.uint32), ref.view(np.uint32)): print("IMAGE MISMATCH", v)
        for k in opts: r.set_option(k, 0) if k.startswith("shadow_order") else None
for v in variants:
    a = np.array(res[v])
    print(f"{v:40s} Mrays/s median {np.median(a[:,0]):8.1f} | ms closest/shadow/shade (min) {a[:,1].min():7.2f} {a[:,2].min():7.2f} {a[:,3].min():7.2f}")
r.close()
