#!/bin/bash
# round 6, task 1: refill cost / idle lanes.  Base library (round 5's kernel) against the round-6 kernel (scalar claim state, one-test classification) with and without kShade's
# 1/direction stream, refill thresholds swept; images compared inside the tool.
OUT=gpurun_out/r06_lanes; mkdir -p $OUT
for rep in 1 2; do
echo "== base (round 5 kernel), pass $rep"
RAYFINDER_AMD_LIB=$PWD/rayfinder_amd/librayfinder_amd_base.so python tools/r06/ab_variants.py 64 "refill_min_deep=22" "refill_min_deep=16" "refill_min_deep=12" "refill_min_deep=8" 2>&1 | grep -v "amdgpu.ids\|RAYFINDER_AMD_LIB"
echo "== round 6 kernel, pass $rep"
python tools/r06/ab_variants.py 64 "inv_stream=0,refill_min_deep=22" "inv_stream=0,refill_min_deep=16" "inv_stream=0,refill_min_deep=12" "inv_stream=0,refill_min_deep=8" "inv_stream=0,refill_min_deep=4" \
   "inv_stream=1,refill_min_deep=22" "inv_stream=1,refill_min_deep=16" "inv_stream=1,refill_min_deep=12" "inv_stream=1,refill_min_deep=8" "inv_stream=1,refill_min_deep=4" "inv_stream=1,refill_min_deep=1" 2>&1 | grep -v "amdgpu.ids"
done 2>&1 | tee $OUT/ab_refill_plain.log
