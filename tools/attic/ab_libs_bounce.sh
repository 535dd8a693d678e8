#!/bin/bash
# As tools/ab_libs.sh, with the per-bounce table of tools/gpu_bounce_sweep.py:  bash tools/ab_libs_bounce.sh <spp> "<variant>" ...
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
SPP=$1; shift
for rep in 1 2; do
  for lib in librayfinder_amd.so librayfinder_amd_exp.so; do
    echo "== $lib (pass $rep)"
    RAYFINDER_AMD_LIB=$REPO/rayfinder_amd/$lib python tools/gpu_bounce_sweep.py $SPP "$@" 2>&1 | grep -v amdgpu.ids
  done
done
