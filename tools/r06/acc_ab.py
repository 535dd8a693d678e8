#!/usr/bin/env python3
"""ms of kAccumulateRuns (rf_stats.ms_accumulate) and of the frame for the library in RAYFINDER_AMD_LIB: atrium 1080p, spp per batch from argv (default 320)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import rayfinder_amd as rf
from rayfinder_amd import scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 320
pt, _ = scenes.atrium()
W, H, b = 1920, 1080, 8
cam = rf.fly_camera(W, H)
r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), 0.25), pt.scene())
r.render(spp); r.synchronize()
best = (1e30, 1e30)
for k in range(3):
    r.set_render_parameters(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), 0.3 + 0.01 * k))
    r.set_timing(True); r.reset_stats()
    t0 = time.perf_counter(); r.render(spp); r.synchronize(); dt = time.perf_counter() - t0
    s = r.stats()
    best = (min(best[0], s["ms_accumulate"]), min(best[1], dt * 1e3))
img = r.read_accumulation()[0]
print(f"{os.path.basename(os.environ.get('RAYFINDER_AMD_LIB', 'default')):34s} accumulate {best[0]:6.3f} ms   frame {best[1]:8.2f} ms   image checksum {int(np.asarray(img).view(np.uint32).astype(np.uint64).sum()):d}")
r.close()
