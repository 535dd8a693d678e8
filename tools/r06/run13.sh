#!/bin/bash
OUT=gpurun_out/r06_lanes; mkdir -p $OUT
export PYTHONPATH=$PWD
python tools/r06/ab_variants.py 64 "uniform_fetch=2" "uniform_fetch=-1" "uniform_fetch=1" "uniform_fetch=0" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_uniform_fetch.log
RF_SCENE_DETAIL=clutter python tools/r06/ab_variants.py 64 "uniform_fetch=2" "uniform_fetch=-1" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_uniform_fetch_clutter.log
RF_SCENE_SCALE=8 python tools/r06/ab_variants.py 16 "uniform_fetch=2" "uniform_fetch=-1" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_uniform_fetch_x8.log
