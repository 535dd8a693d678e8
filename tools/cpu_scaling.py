#!/usr/bin/env python3
"""How the oracle's C-side threaded render scales on this host (bench.py's cpu_baseline uses the best thread count it finds):
   tools/cpu_scaling.py [threads ...]   Duck, 640x480, 16 spp, 4 bounces."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import orc
from conftest import DuckOracle
import rayfinder_amd as rf
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(f): print(f, open(f).read().strip())
d = DuckOracle()
W, H, spp, b = 640, 480, 16, 4
rp = orc.make_render_params(W, H, rf.camera_to_array(rf.fly_camera(W, H)), spp, b, 0.25, rf.aligned_sky_state(rf.make_sky()))
for n in [int(x) for x in sys.argv[1:]] or [1, 8, 16, 32, 64, 128, 256]:
    t = time.time(); img, st, k = orc.render_threads(d.scene, rp, 0, spp, 0, 0, W, H, n, 1); dt = time.time() - t
    print(f"threads {n:4d} (started {k:4d}): {dt:7.3f} s  {(st.closestRays + st.shadowRays) / dt * 1e-6:8.2f} Mrays/s")
