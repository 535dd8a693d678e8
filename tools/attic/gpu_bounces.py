#!/usr/bin/env python3
"""Per-bounce queue occupancy and traversal rate on the atrium (1080p by default)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rayfinder_amd as rf
from rayfinder_amd import scenes

W = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 32
b = int(sys.argv[4]) if len(sys.argv) > 4 else 8
pt, info = scenes.atrium()
r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, rf.fly_camera(W, H), spp, b, rf.make_sky(), 0.25), pt.scene())
r.render(spp); r.synchronize()
r.set_render_parameters(rf.make_render_parameters(W, H, rf.fly_camera(W, H), spp, b, rf.make_sky(), 0.5))
r.set_timing(True); r.reset_stats()
r.render(spp); r.synchronize()
s = r.bounce_stats()
print("bounce  closest_rays  ms   Mrays/s | shadow_rays  ms   Mrays/s")
for i in range(len(s["closest_rays"])):
    c, mc, sh, ms = int(s["closest_rays"][i]), s["ms_closest"][i], int(s["shadow_rays"][i]), s["ms_shadow"][i]
    print(f"{i+1:4d} {c:12d} {mc:8.2f} {c/max(mc,1e-9)*1e-3:8.0f} | {sh:12d} {ms:8.2f} {sh/max(ms,1e-9)*1e-3:8.0f}")
