#!/bin/bash
OUT=gpurun_out/r06_lanes; mkdir -p $OUT
export PYTHONPATH=$PWD
RF_SCENE_DETAIL=clutter python tools/r06/ab_variants.py 64 "leaf_vote=20,refill_min_deep=22" "leaf_vote=24,refill_min_deep=22" "leaf_vote=28,refill_min_deep=22" "leaf_vote=32,refill_min_deep=22" "leaf_vote=20,refill_min_deep=28" "leaf_vote=24,refill_min_deep=28" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_leaf_vote_clutter2.log
RF_SCENE_SCALE=8 python tools/r06/ab_variants.py 16 "leaf_vote=20" "leaf_vote=24" "leaf_vote=28" "leaf_vote=20,refill_min_deep=22" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_leaf_vote_x8_2.log
