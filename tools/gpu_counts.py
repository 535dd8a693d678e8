#!/usr/bin/env python3
"""Counting build on the atrium: loop-trip / lane-utilisation breakdown of kTraceWide (RF_DEBUG_COUNTERS=1)."""
import os, sys
os.environ["RF_DEBUG_COUNTERS"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rayfinder_amd as rf
from rayfinder_amd import scenes
W, H, spp, b = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 8
pt, info = scenes.atrium()
r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, rf.fly_camera(W, H), spp, b, rf.make_sky(), 0.25), pt.scene())
for k, v in (x.split("=") for x in sys.argv[3:]):
    r.set_option(k, int(v))
r.set_counting(True); r.reset_stats()
r.render(spp); r.synchronize()
s = r.stats()
print({k: s[k] for k in ("closest_rays", "shadow_rays", "closest_node_visits", "closest_record_fetches", "closest_triangle_tests", "stack_high_water")})
