#!/usr/bin/env python3
"""How the renderer fares when the camera stands far outside the scene (origins beyond the conservative records' origin bound: 4 R + 1 for a root box
within +-R): Duck at 800x600, 4 bounces, 16 spp, the camera 2 / 10 / 100 / 1000 scene sizes away with the field of view narrowed to keep the duck in frame.
Prints Mrays/s, kernel times and the scalar-redo count per distance, and checks a crop against the oracle."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rayfinder_amd as rf
from oracle import orc
from conftest import oracle_scene_from_pt
pt = rf.PtFormat.from_gltf(os.path.join(ROOT, "tests", "golden", "Duck.glb"))
sc, a = oracle_scene_from_pt(pt)
lo, hi = np.array(a["bvhNodes"][0]["min"][:3]), np.array(a["bvhNodes"][0]["max"][:3])
centre, size = 0.5 * (lo + hi), float(np.max(hi - lo))
W, H, spp, b = 800, 600, 16, 4
for dist in (2.0, 10.0, 100.0, 1000.0):
    eye = centre + np.array([0.6, 0.4, 0.7]) / np.linalg.norm([0.6, 0.4, 0.7]) * dist * size
    vfov = float(2.0 * np.arctan(0.75 * size / (dist * size)))
    cam = rf.create_camera(eye, centre, 0.0, 1.0, vfov, W / H)
    r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), 0.25), pt.scene())
    r.render(spp); r.synchronize()
    r.set_render_parameters(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), 0.5))
    r.set_timing(True); r.reset_stats()
    t0 = time.perf_counter(); r.render(spp); r.synchronize(); dt = time.perf_counter() - t0
    s = r.stats(); img, _ = r.read_accumulation(); r.close()
    x0, y0, x1, y1 = 360, 260, 440, 320
    ref, _ = orc.render(sc, orc.make_render_params(W, H, rf.camera_to_array(cam), spp, b, 0.5, orc.aligned_sky_state()), spp, spp, x0, y0, x1, y1, accumulated_start=0)
    same = np.array_equal(img[y0:y1, x0:x1, :3].view(np.uint32), ref[y0:y1, x0:x1, :3].view(np.uint32))
    print(f"distance {dist:7.1f} x size: {(s['closest_rays'] + s['shadow_rays']) / dt * 1e-6:8.1f} Mrays/s | ms closest {s['ms_closest']:7.2f} shadow {s['ms_shadow']:7.2f} | rays {s['closest_rays'] + s['shadow_rays']} "
          f"scalar redos {s['scalar_redo_rays']} | crop vs oracle: {'bit-identical' if same else 'DIFFERENT'}")
