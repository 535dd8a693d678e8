#!/usr/bin/env python3
"""Soak of the GPU BVH builder: random soups of random sizes / degeneracies, node bytes == host builder."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import rayfinder_amd as rf
count = int(sys.argv[1]) if len(sys.argv) > 1 else 150
bad = 0
signs = 0
for seed in range(count):
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([1, 2, 3, 5, 17, 63, 64, 65, 129, 1000, 5000, 20000, 60000]))
    c = rng.uniform(-10, 10, (n, 1, 3)) * rng.choice([1.0, 0.0, 1e-3], 3)           # flat / collapsed axes now and then
    tris = (c + rng.normal(0, rng.choice([0.0, 1e-4, 0.3]), (n, 3, 3))).astype(np.float32)
    if seed % 3 == 0: tris = tris[rng.integers(0, n, n)]                               # duplicates
    if seed % 5 == 0: tris = np.round(tris * 4) / 4                                    # lattice coordinates: ties everywhere
    tris = np.ascontiguousarray(tris.reshape(-1, 9), np.float32)
    hn, hi, hd = rf.build_bvh(tris)
    gn, gi, gd, ms = rf.build_bvh_gpu(tris)
    ok = hn.tobytes() == gn.tobytes() and hd == gd and np.array_equal(np.sort(np.asarray(gi)), np.arange(len(tris)))
    if not ok:
        # the documented exception: a box coordinate that is zero may differ in SIGN when a node holds both -0.0f and +0.0f
        same_values = len(hn) == len(gn) and all(np.array_equal(hn[f], gn[f]) for f in hn.dtype.names)   # float ==: -0.0 == +0.0
        diff_bits = (hn["min"].view(np.uint32) != gn["min"].view(np.uint32)) | (hn["max"].view(np.uint32) != gn["max"].view(np.uint32)) if same_values else None
        only_zero_signs = same_values and bool(((hn["min"] == 0) | ~diff_bits[..., :3].reshape(hn["min"].shape)).all()) if same_values else False
        if same_values and hd == gd:
            signs += 1
        else:
            bad += 1
            print("seed", seed, "n", n, "MISMATCH", len(hn), len(gn), hd, gd)
print(f"{count} soups: {bad} mismatches, {signs} differing only in the sign of zero coordinates")
