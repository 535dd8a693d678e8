#!/usr/bin/env python3
"""Half-precision quad records on the atrium at several tessellation scales: the surface-area ratio the default rests on
(rf_wide_layout_stats) next to the measured traversal time with and without them:  tools/half_ratio_sweep.py [scales...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import rayfinder_amd as rf
from rayfinder_amd import scenes
W, H, b, spp = 1920, 1080, 8, 32
cam = rf.fly_camera(W, H)
for scale in [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]:
    if scale > 1: rf.set_bake_bvh_builder(0)
    pt, info = scenes.atrium(scale)
    rf.set_bake_bvh_builder(None)
    st = rf.wide_layout_stats(pt.arrays()["bvhNodes"])
    r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), 0.25), pt.scene())
    r.set_timing(True); r.render(spp); r.synchronize()
    res = {}
    layouts = {"exact": dict(quad_half_from_bounce=0, quad_half_shadow_from_bounce=0, quad_local_from_bounce=0, quad_local_shadow_from_bounce=0),
               "half": dict(quad_half_from_bounce=1, quad_half_shadow_from_bounce=1, quad_local_from_bounce=0, quad_local_shadow_from_bounce=0),
               "local": dict(quad_half_from_bounce=0, quad_half_shadow_from_bounce=0, quad_local_from_bounce=1, quad_local_shadow_from_bounce=1)}
    n = 0
    for rd in range(3):
        for name, opts in layouts.items():
            for k, v in opts.items(): r.set_option(k, v)
            n += 1
            r.set_render_parameters(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), 0.25 + 0.01 * n))
            r.set_timing(True); r.reset_stats(); r.render(spp); r.synchronize()
            s = r.stats()
            res.setdefault(name, []).append((s["ms_closest"], s["ms_shadow"]))
    m = {k: np.min(np.array(v[1:]), axis=0) for k, v in res.items()}      # (the first round warms the timers up)
    e = m["exact"]
    print(f"scale {scale}: {info['triangles']} triangles, half/exact box area {st['quad_half_area_ratio']:.4f} | closest exact {e[0]:.2f} ms, half {(m['half'][0] / e[0] - 1) * 100:+.1f} %, local {(m['local'][0] / e[0] - 1) * 100:+.1f} % | "
          f"shadow exact {e[1]:.2f} ms, half {(m['half'][1] / e[1] - 1) * 100:+.1f} %, local {(m['local'][1] / e[1] - 1) * 100:+.1f} %", flush=True)
    r.close(); del r, pt
