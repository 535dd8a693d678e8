#!/bin/bash
# Occluder-cache sweep (round 4): hint level (RF_OCCLUDER_HINT_LEVELS: quad levels above the occluding leaf where the next rays start) x option sets,
# per-bounce launch times:   bash tools/ab_occluder.sh <spp> "<levels> ..." "<variant>" ...
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
SPP=$1; LEVELS=${2:-"0 1 2"}; shift; shift
for L in $LEVELS; do
  echo "== RF_OCCLUDER_HINT_LEVELS=$L"
  RF_OCCLUDER_HINT_LEVELS=$L RF_OPT_DEFAULTS="occluder_cache_bounces=0,occluder_grid_cells=0" python tools/gpu_bounce_sweep.py $SPP "$@" 2>&1 | grep -v amdgpu.ids
done
