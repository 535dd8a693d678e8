#!/usr/bin/env python3
"""Where the lanes of the traversal waves are busy: the phase counters of an RF_EXP_PHASE build, cumulative over bounces 1..b.

  make -C rayfinder_amd/csrc lib EXP=RF_EXP_PHASE LIBNAME=librayfinder_amd_exp.so
  RAYFINDER_AMD_LIB=$PWD/rayfinder_amd/librayfinder_amd_exp.so RF_DEBUG_COUNTERS=1 python tools/gpu_phase.py [spp = 32] [scene scale = 1]

One "[rf-phase]" line per kernel kind and bounce count; differences of consecutive lines are the single bounces.
"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rayfinder_amd as rf
from rayfinder_amd import scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
scale = int(sys.argv[2]) if len(sys.argv) > 2 else 1
opts = sys.argv[3:]
pt, info = scenes.atrium(scale, os.environ.get("RF_SCENE_DETAIL", "plain"))          # RF_SCENE_DETAIL=clutter: the harder stand-in
W, H = 1920, 1080
cam = rf.fly_camera(W, H)
for b in (1, 2, 3, 4, 8):
    r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), 0.25), pt.scene())
    for o in opts:
        k, v = o.split("=")
        r.set_option(k, int(v))
    r.render(spp); r.synchronize()
    r.set_render_parameters(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), 0.5))
    r.reset_stats()
    r.render(spp); r.synchronize()
    print(f"--- bounces 1..{b}", file=sys.stderr, flush=True)
    r.stats()
    r.close()
