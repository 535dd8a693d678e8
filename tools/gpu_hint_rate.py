#!/usr/bin/env python3
"""Share of the shadow rays that kShadowHint answers (occluder grid), per grid resolution and bounce count:
   tools/gpu_hint_rate.py [spp = 64] [cells ...]     (RF_SCENE_DETAIL=clutter, RF_SCENE_SCALE as elsewhere)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rayfinder_amd as rf
from rayfinder_amd import scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cells = [int(c) for c in sys.argv[2:]] or [256, 512, 1024, 2048]
pt, info = scenes.atrium(int(os.environ.get("RF_SCENE_SCALE", 1)), os.environ.get("RF_SCENE_DETAIL", "plain"))
W, H = 1920, 1080
cam = rf.fly_camera(W, H)
for c in cells:
    for b in (1, 2, 8):
        r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), 0.25), pt.scene())
        r.set_option("shadow_hint_bounces", 64); r.set_option("occluder_grid_cells", c)
        line = f"cells {c:5d} bounces 1..{b}:"
        for rep in range(3):
            r.set_render_parameters(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), 0.25 + 0.01 * rep))
            r.set_timing(True); r.reset_stats(); r.render(spp); r.synchronize()
            s = r.stats()
            line += f"  pass {rep}: answered {s['shadow_rays_hint_answered'] / max(s['shadow_rays'], 1):.3f} of {s['shadow_rays'] / 1e6:.1f} M, shadow {s['ms_shadow']:.2f} ms"
        print(line, flush=True)
        r.close()
