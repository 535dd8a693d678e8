#!/usr/bin/env python3
"""Per-layout closest-hit / shadow launch times and scalar-redo counts of one stand-in: tools/gpu_layout_diag.py plain|clutter [scale]
(default layouts, exact quad records, binary records) -- how the stack-overflow cliff of the clutter scene was found in round 4."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import rayfinder_amd as rf
from rayfinder_amd import scenes
scale = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if scale > 1: rf.set_bake_bvh_builder(0)
pt, info = scenes.atrium(scale, sys.argv[1])
W,H,spp,b = 1920,1080,8,8
r = rf.ReferencePathTracer(rf.make_render_parameters(W,H,rf.fly_camera(W,H),spp,b,rf.make_sky(),0.25), pt.scene())
for opts in (dict(), dict(quad_half_from_bounce=0, quad_half_shadow_from_bounce=0, quad_local_from_bounce=0, quad_local_shadow_from_bounce=0), dict(quad_from_bounce=0, quad_shadow_from_bounce=0)):
    for k,v in opts.items(): r.set_option(k,v)
    r.set_render_parameters(rf.make_render_parameters(W,H,rf.fly_camera(W,H),spp,b,rf.make_sky(),0.25+0.01*len(opts)))
    r.set_timing(True); r.reset_stats(); r.render(spp); r.synchronize()
    s = r.stats()
    print(sys.argv[1], opts, {k:s[k] for k in ("closest_rays","shadow_rays","scalar_redo_rays","abandoned_rays","ms_closest","ms_shadow","ms_shade")})
    bs=r.bounce_stats(); print("   ms_closest", [round(x,2) for x in bs["ms_closest"][:8]], "ms_shadow", [round(x,2) for x in bs["ms_shadow"][:8]])
