#!/usr/bin/env python3
"""Within-process interleaved A/B of traversal variants on the atrium (1080p, 8 bounces)."""
import os, sys, time, json
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import rayfinder_amd as rf
from rayfinder_amd import scenes

# variants: "0" = simple kernels; "1:refill:vote" = persistent with parameters
variants = (sys.argv[1] if len(sys.argv) > 1 else "0,1").split(",")
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 32
pt, info = scenes.atrium()
W, H, b = 1920, 1080, 8
cam = rf.fly_camera(W, H)
r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), 0.25), pt.scene())
r.render(4); r.synchronize()
ref_img = None
res = {v: [] for v in variants}
expo = 1.0
for rd in range(rounds):
    for v in variants:
        parts = [int(x) for x in v.split(":")]
        r.set_option("traversal_variant", parts[0])
        if len(parts) > 1: r.set_option("refill_min", parts[1])
        if len(parts) > 2: r.set_option("leaf_vote", parts[2])
        if len(parts) > 3: r.set_option("chunk", parts[3])
        if len(parts) > 4: r.set_option("shadow_nearest_first", parts[4])
        expo *= 0.99
        r.set_render_parameters(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), expo))
        # frameCount must be a multiple of spp for identical sample sets: pad
        r.set_timing(True); r.reset_stats()
        t0 = time.perf_counter(); r.render(spp); r.synchronize(); dt = time.perf_counter() - t0
        s = r.stats()
        rays = s["closest_rays"] + s["shadow_rays"]
        res[v].append(dict(mrays=rays / dt * 1e-6, ms_closest=s["ms_closest"], ms_shadow=s["ms_shadow"], ms_shade=s["ms_shade"], ms_raygen=s["ms_raygen"]))
        img, _ = r.read_accumulation()
        if ref_img is None:
            ref_img = img
        else:
            same = np.array_equal(img.view(np.uint32), ref_img.view(np.uint32))
            if not same:
                print("IMAGE MISMATCH variant", v, "max abs", float(np.abs(img - ref_img).max()))
for v in variants:
    m = res[v]
    print("variant", v, "Mrays/s median", round(float(np.median([x["mrays"] for x in m])), 1), "best", round(max(x["mrays"] for x in m), 1),
          "| ms closest/shadow/shade/raygen (min):", *(round(min(x[k] for x in m), 2) for k in ("ms_closest", "ms_shadow", "ms_shade", "ms_raygen")))
