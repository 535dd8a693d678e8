#!/bin/bash
# round 6: leaf_vote / chunk re-swept on the round-6 kernel (deep refill threshold 12)
OUT=gpurun_out/r06_lanes; mkdir -p $OUT
export PYTHONPATH=$PWD
python tools/r06/ab_variants.py 64 "leaf_vote=20" "leaf_vote=12" "leaf_vote=16" "leaf_vote=24" "leaf_vote=28" "leaf_vote=32" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_leaf_vote.log
python tools/r06/ab_variants.py 64 "leaf_vote=20,chunk=128,chunk_early=256" "leaf_vote=20,chunk=64,chunk_early=256" "leaf_vote=20,chunk=256,chunk_early=256" "leaf_vote=20,chunk=128,chunk_early=512" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_chunk.log
python -m pytest tests/test_perf.py -q -m perf 2>&1 | tail -3 | tee $OUT/pytest_perf.log
