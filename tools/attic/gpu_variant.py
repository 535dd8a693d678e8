#!/usr/bin/env python3
"""One frame of the atrium per option set, in one process (for counter passes: the kernel instantiations have distinct names):
   tools/gpu_variant.py spp "name=value,name=value" "name=value" ...   ('-' = no option)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rayfinder_amd as rf
from rayfinder_amd import scenes
spp = int(sys.argv[1]); variants = sys.argv[2:] or ["-"]
pt, info = scenes.atrium()
W, H, b = 1920, 1080, 8
cam = rf.fly_camera(W, H)
r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), 0.25), pt.scene())
expo = 1.0
for v in variants:
    if v != "-":
        for kv in v.split(","):
            k, val = kv.split("=")
            r.set_option(k, int(val))
    expo *= 0.99
    r.set_render_parameters(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), expo))     # restarts the accumulation
    r.set_timing(True); r.reset_stats()
    r.render(spp); r.synchronize()
    s = r.stats()
    print(v, "ms closest/shadow", round(s["ms_closest"], 2), round(s["ms_shadow"], 2))
r.close()
