#!/usr/bin/env python3
"""Two frames of the atrium (1080p, 8 bounces) under ONE option set -- a warm-up frame and a measured one -- and a JSON line with
the measured frame's rays and HIP-event milliseconds per bounce (for tools/pmc_bounces.sh):
   tools/gpu_variant_bounces.py spp "name=value,name=value"   ('-' = defaults)"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rayfinder_amd as rf
from rayfinder_amd import scenes
spp = int(sys.argv[1]); v = sys.argv[2] if len(sys.argv) > 2 else "-"
if int(os.environ.get("RF_SCENE_SCALE", 1)) > 1: rf.set_bake_bvh_builder(0)      # GPU builder: same node bytes, 40x faster at that size
pt, info = scenes.atrium(int(os.environ.get("RF_SCENE_SCALE", 1)), os.environ.get("RF_SCENE_DETAIL", "plain"))      # RF_SCENE_DETAIL=clutter: the harder stand-in
W, H, b = 1920, 1080, int(os.environ.get("RF_B", 8))
cam = rf.fly_camera(W, H)
r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), 0.25), pt.scene())
if v != "-":
    for kv in v.split(","):
        k, val = kv.split("=")
        r.set_option(k, int(val))
for expo in (0.99, 0.98):
    r.set_render_parameters(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), expo))     # restarts the accumulation
    r.set_timing(True); r.reset_stats()
    r.render(spp); r.synchronize()
s = r.bounce_stats()
print(json.dumps(dict(variant=v, closest=[int(x) for x in s["closest_rays"]], shadow=[int(x) for x in s["shadow_rays"]],
                      ms_closest=[float(x) for x in s["ms_closest"]], ms_shadow=[float(x) for x in s["ms_shadow"]])))
r.close()
