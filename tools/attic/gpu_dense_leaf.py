#!/usr/bin/env python3
"""Round 5: the dense (lane, triangle) leaf phase on the stand-ins -- per-bounce launch times for dense_leaf_min = 0 (off) / 2 / 3 / 4, interleaved, images compared bit for bit.
   RF_SCENE_DETAIL=clutter RF_SCENE_SCALE=1 python tools/gpu_dense_leaf.py <spp> [variants...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import rayfinder_amd as rf
from rayfinder_amd import scenes
spp = int(sys.argv[1]); variants = sys.argv[2:] or ["dense_leaf_min=0", "dense_leaf_min=2", "dense_leaf_min=3", "dense_leaf_min=4"]
if int(os.environ.get("RF_SCENE_SCALE", 1)) > 1: rf.set_bake_bvh_builder(0)
pt, info = scenes.atrium(int(os.environ.get("RF_SCENE_SCALE", 1)), os.environ.get("RF_SCENE_DETAIL", "plain"))
W, H, b = 1920, 1080, 8
cam = rf.fly_camera(W, H)
r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), 0.25), pt.scene())
r.render(spp); r.synchronize()
ref = None; expo = 1.0; res = {v: [] for v in variants}
for rd in range(3):
    for v in variants:
        for kv in v.split(","):
            k, val = kv.split("="); r.set_option(k, int(val))
        expo *= 0.99
        r.set_render_parameters(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), expo))
        r.set_timing(True); r.reset_stats()
        r.render(spp); r.synchronize()
        s = r.bounce_stats()
        res[v].append(np.concatenate([np.asarray(s["ms_closest"], float), np.asarray(s["ms_shadow"], float)]))
        img, _ = r.read_accumulation()
        if ref is None: ref = img
        elif not np.array_equal(img.view(np.uint32), ref.view(np.uint32)): print("IMAGE MISMATCH", v, int((img.view(np.uint32) != ref.view(np.uint32)).sum()))
print(info["name"])
print(f"{'':34s}" + " ".join(f"  c{i+1:<4d}" for i in range(b)) + " | " + " ".join(f"  s{i+1:<4d}" for i in range(b)) + " | closest  shadow")
for v in variants:
    m = np.min(np.array(res[v]), axis=0)
    print(f"{v:34s}" + " ".join(f"{x:7.2f}" for x in m[:b]) + " | " + " ".join(f"{x:7.2f}" for x in m[b:]) + f" | {m[:b].sum():7.2f} {m[b:].sum():7.2f}")
r.close()
