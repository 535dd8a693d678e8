#!/usr/bin/env python3
"""tools/gpu_opt2.py on a chosen stand-in: RF_SCENE_DETAIL=clutter python tools/gpu_opt_scene.py <spp> [variants...]  (see gpu_opt2.py)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import rayfinder_amd as rf
from rayfinder_amd import scenes
spp = int(sys.argv[1]); variants = sys.argv[2:] or ["-"]
pt, info = scenes.atrium(int(os.environ.get("RF_SCENE_SCALE", 1)), os.environ.get("RF_SCENE_DETAIL", "plain"))
W, H, b = 1920, 1080, 8
cam = rf.fly_camera(W, H)
r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), 0.25), pt.scene())
r.render(spp); r.synchronize()
ref = None; expo = 1.0; res = {v: [] for v in variants}
for rd in range(3):
    for v in variants:
        if v != "-":
            for kv in v.split(","):
                k, val = kv.split("="); r.set_option(k, int(val))
        expo *= 0.99
        r.set_render_parameters(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), expo))
        r.set_timing(True); r.reset_stats()
        t0 = time.perf_counter(); r.render(spp); r.synchronize(); dt = time.perf_counter() - t0
        s = r.stats()
        res[v].append(((s["closest_rays"] + s["shadow_rays"]) / dt * 1e-6, s["ms_closest"], s["ms_shadow"], s["ms_shade"], s["scalar_redo_rays"]))
        img, _ = r.read_accumulation()
        if ref is None: ref = img
        elif not np.array_equal(img.view(np.uint32), ref.view(np.uint32)): print("IMAGE MISMATCH", v)
for v in variants:
    a = np.array(res[v])
    print(f"{info['name'][:34]:34s} {v:30s} Mrays/s median {np.median(a[:,0]):8.1f} | ms closest/shadow/shade (min) {a[:,1].min():7.2f} {a[:,2].min():7.2f} {a[:,3].min():7.2f} | scalar redos {int(a[:,4].max())}")
r.close()
