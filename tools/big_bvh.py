import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np, rayfinder_amd as rf
rng = np.random.default_rng(3)
n = 4_000_000
c = rng.uniform(-50, 50, (n, 1, 3)).astype(np.float32)
tris = (c + rng.normal(0, 0.2, (n, 3, 3)).astype(np.float32)).reshape(-1, 9).astype(np.float32)
t0 = time.time(); hn, hi, hd = rf.build_bvh(tris); th = time.time() - t0
gn, gi, gd, ms = rf.build_bvh_gpu(tris)
gn2, gi2, gd2, ms2 = rf.build_bvh_gpu(tris)
print(f"{n} tris: host {th*1e3:.0f} ms, gpu {ms:.1f} / {ms2:.1f} ms, nodes {len(hn)} {len(gn)}, identical {hn.tobytes() == gn.tobytes()}, depth {hd} {gd}, deterministic {np.array_equal(gi, gi2)}")
