#!/usr/bin/env python3
"""Per-bounce traversal times on the atrium for several option sets, interleaved in one process:
   tools/gpu_bounce_sweep.py spp "name=value,..." ...   ('-' = defaults; RF_OPT_DEFAULTS as in gpu_opt2.py)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import rayfinder_amd as rf
from rayfinder_amd import scenes
spp = int(sys.argv[1]); variants = sys.argv[2:] or ["-"]
defaults = dict(kv.split("=") for kv in os.environ.get("RF_OPT_DEFAULTS", "").split(",") if kv)
if os.environ.get("RF_SCENE", "atrium") == "duck":      # BASELINE.json config 2
    pt = rf.PtFormat.from_gltf(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "Duck.glb"))
    W, H, b = 800, 600, 4
else:
    if int(os.environ.get("RF_SCENE_SCALE", 1)) > 1: rf.set_bake_bvh_builder(0)      # GPU builder: same node bytes, 40x faster at that size
    pt, info = scenes.atrium(int(os.environ.get("RF_SCENE_SCALE", 1)), os.environ.get("RF_SCENE_DETAIL", "plain"))
    W, H, b = 1920, 1080, 8
cam = rf.fly_camera(W, H)
r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), 0.25), pt.scene())
r.render(spp); r.synchronize()
expo = 1.0; res = {v: [] for v in variants}
for rd in range(3):
    for v in variants:
        for k, val in defaults.items(): r.set_option(k, int(val))
        if v != "-":
            for kv in v.split(","):
                k, val = kv.split("="); r.set_option(k, int(val))
        expo *= 0.99
        r.set_render_parameters(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), expo))
        r.set_timing(True); r.reset_stats()
        r.render(spp); r.synchronize()
        s = r.bounce_stats()
        res[v].append(np.concatenate([np.asarray(s["ms_closest"], float), np.asarray(s["ms_shadow"], float)]))
print(f"{'':44s}" + " ".join(f"  c{i+1:<4d}" for i in range(b)) + " | " + " ".join(f"  s{i+1:<4d}" for i in range(b)) + " |  total")
for v in variants:
    m = np.min(np.array(res[v]), axis=0)
    print(f"{v:44s}" + " ".join(f"{x:7.2f}" for x in m[:b]) + " | " + " ".join(f"{x:7.2f}" for x in m[b:]) + f" | {m.sum():7.2f}")
r.close()
