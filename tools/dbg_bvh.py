import sys; sys.path.insert(0, '.')
import numpy as np, rayfinder_amd as rf
rng = np.random.default_rng(1)
tris = rng.uniform(-1, 1, (5000, 9)).astype(np.float32)
try:
    out = rf.build_bvh_gpu(tris); print("first ok", len(out[0]), out[3])
except Exception as e:
    print("first failed:", e)
import torch; print(torch.cuda.is_available())
out = rf.build_bvh_gpu(tris); print("second ok", len(out[0]), out[2], out[3])
h = rf.build_bvh(tris)
print("nodes equal:", out[0].tobytes() == h[0].tobytes(), len(h[0]), h[2])
