#!/usr/bin/env python3
"""Soak version of tests/test_gpu_parity.py::test_random_scenes_cameras_and_skies_bit_identical_to_oracle: 200 more seeds."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.chdir(ROOT)
import test_gpu_parity as T
first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 12), (int(sys.argv[2]) if len(sys.argv) > 2 else 200)
bad = 0
for seed in range(first, first + count):
    try:
        T.test_random_scenes_cameras_and_skies_bit_identical_to_oracle(seed)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED", str(e)[:300])
print(f"seeds {first}..{first + count - 1}: {bad} failures")
