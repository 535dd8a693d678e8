#!/usr/bin/env python3
"""Ad-hoc GPU probe on the synthetic atrium: image dump, oracle parity on a crop, timing."""
import os, sys, time, json
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import rayfinder_amd as rf
from rayfinder_amd import scenes
from oracle import orc

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
t0 = time.time(); pt, info = scenes.atrium(); print(info, "gen+bvh s", time.time() - t0)
a = pt.arrays()
W, H, spp, b = 1920, 1080, 16, 8
cam = rf.fly_camera(W, H)
params = rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), 0.25)
r = rf.ReferencePathTracer(params, pt.scene())
r.render(spp); r.synchronize()
img, acc = r.read_accumulation()
print("mean radiance", img[..., :3].mean() / acc, "nan px", int(np.isnan(img[..., :3]).any(axis=-1).sum()))
bgra = r.read_tonemapped()
from PIL import Image
rgb = np.stack([(bgra >> 16) & 255, (bgra >> 8) & 255, bgra & 255], axis=-1).astype(np.uint8)
Image.fromarray(rgb).resize((960, 540)).save(os.path.join(ROOT, "gpurun_out", "atrium.png"))

# oracle parity on crops
descs = []; off = 0
for (px, w, h) in a["baseColorTextures"]:
    descs.append((w, h, off)); off += px.size
texels = np.concatenate([px for (px, _, _) in a["baseColorTextures"]])
sc = orc.OracleScene(a["bvhNodes"], a["trianglePositionAttributes"], a["triangleVertexAttributes"], np.array(descs, np.uint32), texels)
rp = orc.make_render_params(W, H, rf.camera_to_array(cam), spp, b, 0.25, rf.aligned_sky_state(rf.make_sky()))
for (x0, y0) in [(928, 508), (200, 800), (1500, 300)]:
    t0 = time.time()
    ref, st = orc.render(sc, rp, 0, spp, x0, y0, x0 + 32, y0 + 32)
    g = img[y0:y0 + 32, x0:x0 + 32, :3]; c = ref[y0:y0 + 32, x0:x0 + 32, :3]
    print("crop", (x0, y0), "oracle s", round(time.time() - t0, 2), "bit-identical frac", float((g == c).mean()), "max abs", float(np.abs(g - c).max()),
          "stackHigh", st.stackHigh, "oob", st.texelOobClamps, "mean", float(c.mean()))

# timing
spp = 16
for trial in range(2):
    r.set_render_parameters(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), 0.5 + trial))
    r.reset_stats()
    t0 = time.time(); r.render(spp); r.synchronize(); dt = time.time() - t0
    s = r.stats(); rays = s["closest_rays"] + s["shadow_rays"]
    print(json.dumps(dict(trial=trial, seconds=dt, mrays=rays / dt * 1e-6, closest=s["closest_rays"], shadow=s["shadow_rays"])))
r.set_render_parameters(params); r.set_counting(True); r.set_timing(True); r.reset_stats()
r.render(spp); r.synchronize(); s = r.stats()
print(json.dumps(s))
cv, sv = s["closest_node_visits"], s["shadow_node_visits"]
print("closest: visits/ray", cv / s["closest_rays"], "tri/ray", s["closest_triangle_tests"] / s["closest_rays"],
      "alg GB/s", (s["closest_rays"] * 44 + 48 * (cv + s["closest_triangle_tests"])) / (s["ms_closest"] * 1e-3) / 1e9)
print("shadow: visits/ray", sv / s["shadow_rays"], "tri/ray", s["shadow_triangle_tests"] / s["shadow_rays"],
      "alg GB/s", (s["shadow_rays"] * 32 + 48 * (sv + s["shadow_triangle_tests"])) / (s["ms_shadow"] * 1e-3) / 1e9)
