#!/usr/bin/env python3
"""Ad-hoc GPU probe: smoke + timing of Duck at config-2 size. Not part of the test suite."""
import os, sys, time, json
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
import rayfinder_amd as rf

g.smoke()
pt = rf.PtFormat.from_gltf(os.path.join(ROOT, "tests", "golden", "Duck.glb"))
for (W, H, spp, b) in [(800, 600, 64, 4), (1920, 1080, 32, 8)]:
    params = rf.make_render_parameters(W, H, rf.fly_camera(W, H), spp, b, rf.make_sky(), 0.25)
    r = rf.ReferencePathTracer(params, pt.scene())
    r.render(spp); r.synchronize()  # warm
    r.set_render_parameters(rf.make_render_parameters(W, H, rf.fly_camera(W, H), spp, b, rf.make_sky(), 0.5))
    r.reset_stats()
    t0 = time.time(); r.render(spp); r.synchronize(); dt = time.time() - t0
    s = r.stats()
    rays = s["closest_rays"] + s["shadow_rays"]
    print(json.dumps(dict(W=W, H=H, spp=spp, bounces=b, seconds=dt, mrays=rays / dt * 1e-6, avg_pass_ms=r.average_renderpass_duration_ms(), **s)))
    # counting + timing pass
    r.set_render_parameters(rf.make_render_parameters(W, H, rf.fly_camera(W, H), spp, b, rf.make_sky(), 0.25))
    r.set_counting(True); r.set_timing(True); r.reset_stats()
    r.render(spp); r.synchronize()
    s = r.stats()
    print(json.dumps(s))
    img, acc = r.read_accumulation()
    print("mean radiance", img[..., :3].mean() / acc, "nan", int(np.isnan(img).sum()))
    r.close()
