#!/bin/bash
# round 6: what the lanes that take no step are doing -- RF_EXP_PHASE build, deep refill threshold 22 / 12 / 1 (cumulative over bounces 1..b; differences of lines = single bounces)
OUT=gpurun_out/r06_lanes; mkdir -p $OUT
export PYTHONPATH=$PWD
for t in 22 12 1; do
  echo "===== refill_min_deep=$t"
  RAYFINDER_AMD_LIB=$PWD/rayfinder_amd/librayfinder_amd_phase.so RF_DEBUG_COUNTERS=1 python tools/gpu_phase.py 16 1 refill_min_deep=$t 2>&1 | grep -E "^---|rf-phase\] closest"
done > $OUT/phase_lanes.log 2>&1
cat $OUT/phase_lanes.log
