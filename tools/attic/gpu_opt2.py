#!/usr/bin/env python3
"""Interleaved A/B of renderer options on the atrium with explicit values per variant:
   tools/gpu_opt2.py spp "name=value,name=value" "name=value" ...   ('-' = defaults; every option named anywhere is reset to its
   default value, given as name:default in RF_OPT_DEFAULTS="name=default,...", before a variant is applied)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import rayfinder_amd as rf
from rayfinder_amd import scenes
spp = int(sys.argv[1]); variants = sys.argv[2:] or ["-"]
defaults = dict(kv.split("=") for kv in os.environ.get("RF_OPT_DEFAULTS", "").split(",") if kv)
rounds = 3
if int(os.environ.get("RF_SCENE_SCALE", 1)) > 1: rf.set_bake_bvh_builder(0)      # GPU builder: same node bytes, 40x faster at that size
pt, info = scenes.atrium(scale=int(os.environ.get("RF_SCENE_SCALE", 1)))
W, H, b = int(os.environ.get("RF_W", 1920)), int(os.environ.get("RF_H", 1080)), int(os.environ.get("RF_B", 8))
cam = rf.fly_camera(W, H)
r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), 0.25), pt.scene())
r.render(spp); r.synchronize()
ref = None; expo = 1.0; res = {v: [] for v in variants}
for rd in range(rounds):
    for v in variants:
        for k, val in defaults.items(): r.set_option(k, int(val))
        opts = dict(kv.split("=") for kv in v.split(",")) if v != "-" else {}
        for k, val in opts.items(): r.set_option(k, int(val))
        expo *= 0.99
        r.set_render_parameters(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), expo))
        r.set_timing(True); r.reset_stats()
        t0 = time.perf_counter(); r.render(spp); r.synchronize(); dt = time.perf_counter() - t0
        s = r.stats()
        res[v].append(((s["closest_rays"] + s["shadow_rays"]) / dt * 1e-6, s["ms_closest"], s["ms_shadow"], s["ms_shade"], s["ms_raygen"], s["ms_accumulate"]))
        img, _ = r.read_accumulation()
        if ref is None: ref = img
        elif not np.array_equal(img.view(np.uint32), ref.view(np.uint32)): print("IMAGE MISMATCH", v)
for v in variants:
    a = np.array(res[v])
    print(f"{v:40s} Mrays/s median {np.median(a[:,0]):8.1f} best {a[:,0].max():8.1f} | ms closest/shadow/shade/raygen/accumulate (min) {a[:,1].min():7.2f} {a[:,2].min():7.2f} {a[:,3].min():7.2f} {a[:,4].min():6.2f} {a[:,5].min():6.2f}")
r.close()
