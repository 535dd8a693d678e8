#!/usr/bin/env python3
"""One seed of the randomized differential test with explicit renderer options (bisecting a mismatch):
   tools/gpu_fuzz_one.py <seed> ["name=value,name=value" ...]   each option set is rendered and compared with the oracle image"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); os.chdir(ROOT)
import numpy as np
import rayfinder_amd as rf
from oracle import orc
import test_gpu_parity as T
from conftest import oracle_scene_from_pt, bits
seed = int(sys.argv[1]); variants = sys.argv[2:] or ["-"]
rng = np.random.default_rng(1000 + seed)
kind = ["boxes", "slivers", "duplicates", "clutter"][seed % 4]
pt = T._random_scene(rng, kind)
W, H = int(rng.integers(5, 13)) * 8 - int(rng.integers(0, 7)), int(rng.integers(4, 10)) * 8 - int(rng.integers(0, 7))
spp, bounces = int(rng.integers(1, 6)), int(rng.integers(1, 8))
pos = rng.uniform(-3, 3, 3); pos[1] = abs(pos[1]) + 0.3
cam = rf.create_camera(pos, rng.uniform(-1, 1, 3), float(rng.choice([0.0, 0.05, 0.3])), float(rng.uniform(1, 6)), orc.degrees_to_radians(float(rng.uniform(30, 100))), W / H)
sky = rf.make_sky(turbidity=float(rng.uniform(1, 10)), albedo=tuple(rng.uniform(0, 1, 3)), sun_zenith_degrees=float(rng.uniform(0, 89)), sun_azimuth_degrees=float(rng.uniform(0, 360)))
sc, _ = oracle_scene_from_pt(pt)
rp = orc.make_render_params(W, H, rf.camera_to_array(cam), spp, bounces, 0.5, rf.aligned_sky_state(sky))
with np.errstate(all="ignore"):
    ref, st = orc.render(sc, rp, 0, spp)
print("seed", seed, kind, W, H, "spp", spp, "bounces", bounces, "layout stats", rf.wide_layout_stats(pt.arrays()["bvhNodes"]), "oracle", st.as_dict())
for v in variants:
    r, params = T._renderer(pt, W, H, spp, bounces, cam=cam, sky=sky, exposure=0.5)
    if v != "-":
        for kv in v.split(","):
            k, val = kv.split("="); r.set_option(k, int(val))
    r.render(spp)
    img, _ = r.read_accumulation()
    g, c = img[..., :3], ref[..., :3]
    bad = np.argwhere(~((bits(g) == bits(c)) | np.isnan(g)).all(axis=-1))
    print(f"{v:70s} mismatching pixels {len(bad)}", [(int(y), int(x), g[y, x].tolist(), c[y, x].tolist()) for y, x in bad[:3]], {k: val for k, val in r.stats().items() if "redo" in k or "aband" in k})
    r.close()
