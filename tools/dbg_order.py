import sys; sys.path.insert(0, '.')
import numpy as np
import torch
print("torch first:", torch.cuda.is_available())
import rayfinder_amd as rf
rng = np.random.default_rng(1)
tris = rng.uniform(-1, 1, (500, 9)).astype(np.float32)
try:
    out = rf.build_bvh_gpu(tris); print("gpu build ok", len(out[0]))
except Exception as e:
    print("gpu build failed:", e)
pt = rf.PtFormat.from_gltf("tests/golden/Duck.glb")
try:
    r = rf.ReferencePathTracer(rf.make_render_parameters(64, 64, rf.fly_camera(64, 64), 1, 1, rf.make_sky(), 0.25), pt.scene()); print("renderer ok")
except Exception as e:
    print("renderer failed:", e)
