#!/usr/bin/env python3
"""Round 6 A/B driver: option sets interleaved in ONE process on one of the stand-ins, per-bounce closest-hit / shadow launch times (HIP events), kShade + kSky time,
Mrays/s of the frame, and the image of every variant compared bit for bit with the first variant's.
   RF_SCENE_DETAIL=plain|clutter RF_SCENE_SCALE=1|8 python tools/r06/ab_variants.py spp "name=value,..." ...      ('-' = defaults; an option the library does not know skips the variant)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import rayfinder_amd as rf
from rayfinder_amd import scenes
spp = int(sys.argv[1]); variants = sys.argv[2:] or ["-"]
scale = int(os.environ.get("RF_SCENE_SCALE", 1))
if scale > 1: rf.set_bake_bvh_builder(0)
pt, info = scenes.atrium(scale, os.environ.get("RF_SCENE_DETAIL", "plain"))
W, H, b = 1920, 1080, 8
cam = rf.fly_camera(W, H)
rounds = int(os.environ.get("RF_AB_ROUNDS", 3))
res = {v: [] for v in variants}; images = {}; skipped = set()
r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), 0.25), pt.scene())
for rd in range(rounds + 1):                      # round 0: warm-up (allocations, occluder grid) + the image, not timed
    for v in variants:
        if v in skipped: continue
        try:                                      # ONE renderer: every variant names all the options the set varies (an option stays as the last variant left it)
            if v != "-":
                for kv in v.split(","):
                    k, val = kv.split("="); r.set_option(k, int(val))
        except rf.RayfinderError as e:
            print(f"{v}: skipped ({str(e)[:80]})"); skipped.add(v); continue
        r.set_render_parameters(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), 0.25 * (1.0 + 0.001 * (rd * 64 + variants.index(v) + 1))))   # (a new exposure: restarts the accumulation)
        r.set_timing(True); r.reset_stats()
        t0 = time.perf_counter(); r.render(spp); r.synchronize(); wall = time.perf_counter() - t0
        s = r.bounce_stats(); st = r.stats()
        if rd == 0:
            images[v] = r.read_accumulation()[0]
            continue
        rays = st["closest_rays"] + st["shadow_rays"]
        res[v].append(np.concatenate([np.asarray(s["ms_closest"], float)[:b], np.asarray(s["ms_shadow"], float)[:b], [st.get("ms_shade", 0.0), st.get("ms_raygen", 0.0), wall * 1e3, rays / wall / 1e6]]))
first = next(v for v in variants if v not in skipped)
print(f"{'':52s}" + " ".join(f"  c{i+1:<4d}" for i in range(b)) + " |  closest  shadow   shade  raygen |  wall ms   Mrays/s  image")
for v in variants:
    if v in skipped: continue
    m = np.array(res[v]); lo = m.min(axis=0); best = m[:, -1].max()
    same = np.array_equal(images[v].view(np.uint32), images[first].view(np.uint32))
    print(f"{v:52s}" + " ".join(f"{x:7.2f}" for x in lo[:b]) + f" | {lo[:b].sum():8.2f} {lo[b:2*b].sum():7.2f} {lo[2*b]:7.2f} {lo[2*b+1]:7.2f} | {lo[2*b+2]:8.2f} {best:9.1f}  {'identical' if same else 'DIFFERENT'}")
r.close()
