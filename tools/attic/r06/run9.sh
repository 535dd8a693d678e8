#!/bin/bash
OUT=gpurun_out/r06_lanes; mkdir -p $OUT
export PYTHONPATH=$PWD
python tools/r06/ab_variants.py 64 "leaf_vote=20" "leaf_vote=8" "leaf_vote=10" "leaf_vote=12" "leaf_vote=14" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_leaf_vote2.log
RF_SCENE_DETAIL=clutter python tools/r06/ab_variants.py 64 "leaf_vote=20" "leaf_vote=12" "leaf_vote=16" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_leaf_vote_clutter.log
RF_SCENE_SCALE=8 python tools/r06/ab_variants.py 16 "leaf_vote=20" "leaf_vote=12" "leaf_vote=16" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_leaf_vote_x8.log
RF_SCENE=duck python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_leaf_vote_duck.log
import os, sys, time, numpy as np
sys.argv = ["x", "64", "leaf_vote=20", "leaf_vote=12", "leaf_vote=16"]
import rayfinder_amd as rf
pt = rf.PtFormat.from_gltf("tests/golden/Duck.glb")
W, H, b, spp = 800, 600, 4, 64
cam = rf.fly_camera(W, H)
r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), 0.25), pt.scene())
res = {}
for rd in range(6):
    for v in (20, 12, 16):
        r.set_option("leaf_vote", v)
        r.set_render_parameters(rf.make_render_parameters(W, H, cam, spp, b, rf.make_sky(), 0.25 + 0.001 * (rd * 8 + v)))
        r.set_timing(True); r.reset_stats()
        t0 = time.perf_counter(); r.render(spp); r.synchronize(); dt = time.perf_counter() - t0
        s = r.stats()
        if rd: res.setdefault(v, []).append((s["ms_closest"], dt * 1e3))
for v, xs in res.items(): print("duck leaf_vote", v, "closest ms", min(x[0] for x in xs), "wall ms", min(x[1] for x in xs))
PY
