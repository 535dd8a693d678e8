#!/usr/bin/env python3
"""kShade's own-triangle test of the shadow rays (option shadow_self_test): interleaved off / on / off / on, images compared bit for bit, how many shadow rays it settles.
   usage: gpu_self_shadow.py [spp = 64]     (RF_SCENE_DETAIL=clutter, RF_SCENE_SCALE=8 as for the other tools)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import rayfinder_amd as rf
from rayfinder_amd import scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
scale = int(os.environ.get("RF_SCENE_SCALE", 1))
if scale > 1: rf.set_bake_bvh_builder(0)
pt, info = scenes.atrium(scale, os.environ.get("RF_SCENE_DETAIL", "plain"))
W, H, b = 1920, 1080, 8
cam = rf.fly_camera(W, H)
r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), 0.25), pt.scene())
r.render(spp); r.synchronize()
ref = None; expo = 1.0
rows = {0: [], 1: []}
for rd in range(3):
    for on in (0, 1, 0, 1):
        r.set_option("shadow_self_test", on)
        expo *= 0.99
        r.set_render_parameters(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), expo))
        r.set_timing(True); r.reset_stats()
        t0 = time.perf_counter(); r.render(spp); r.synchronize(); dt = time.perf_counter() - t0
        s = r.stats()
        rows[on].append(((s["closest_rays"] + s["shadow_rays"]) / dt * 1e-6, s["ms_closest"], s["ms_shadow"], s["ms_shade"], s["shadow_rays"], s["shadow_rays_self_answered"], s["shadow_rays_hint_answered"]))
        img, _ = r.read_accumulation()
        if ref is None: ref = img
        elif not np.array_equal(img.view(np.uint32), ref.view(np.uint32)): print("IMAGE MISMATCH", on)
for on in (0, 1):
    a = np.array(rows[on], dtype=np.float64)
    print(f"shadow_self_test={on}: Mrays/s median {np.median(a[:,0]):8.1f} | ms closest/shadow/shade (min) {a[:,1].min():7.2f} {a[:,2].min():7.2f} {a[:,3].min():7.2f} | shadow rays {a[0,4]:.0f}, settled by kShade {a[0,5]:.0f} ({a[0,5]/a[0,4]:.3f}), by the first look {a[0,6]:.0f}")
r.close()
