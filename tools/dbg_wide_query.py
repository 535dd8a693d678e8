import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import rayfinder_amd as rf
from oracle import orc
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_gpu_parity as T
pt = rf.PtFormat.from_gltf("tests/golden/Duck.glb")
a = pt.arrays()
nodes = a["bvhNodes"]; pos48 = a["trianglePositionAttributes"]
rng = np.random.default_rng(44)
lo, hi = nodes[0]["min"].astype(np.float64), nodes[0]["max"].astype(np.float64)
rays = T._random_rays(rng, 30000, lo, hi)
r, _ = T._renderer(pt, 64, 64, 1, 1)
r.set_option("query_variant", 2)
tmax = 10000.0
with np.errstate(all="ignore"):
    cpu = orc.intersect_bvh_batch(nodes, pos48, rays, tmax)
gpu = r.intersect_rays(rays, tmax)
for k in ("hit", "tri", "t", "uv", "p"):
    g = gpu[k].view(np.uint32) if gpu[k].dtype == np.float32 else gpu[k]
    c = cpu[k].view(np.uint32) if cpu[k].dtype == np.float32 else cpu[k]
    bad = np.nonzero((g != c).reshape(len(rays), -1).any(axis=1))[0]
    print(k, "mismatches", len(bad), bad[:10])
    for i in bad[:5]:
        print("   ray", rays[i], "gpu", gpu[k][i], "cpu", cpu[k][i], "tri", gpu["tri"][i], cpu["tri"][i])
