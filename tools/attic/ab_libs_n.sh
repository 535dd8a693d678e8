#!/bin/bash
# A/B/C... of several builds of the library, interleaved in one gpurun call (round 4; ab_libs.sh is the two-library form):
#   LIBS="librayfinder_amd.so librayfinder_amd_a.so librayfinder_amd_b.so" bash tools/ab_libs_n.sh <spp> "<variant>" ...
#   (variants as for tools/gpu_opt2.py; RF_AB_TOOL=gpu_bounce_sweep.py prints per-bounce launch times instead)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
SPP=$1; shift
TOOL=${RF_AB_TOOL:-gpu_opt2.py}
for rep in 1 2; do
  for lib in $LIBS; do
    echo "== $lib (pass $rep)"
    RAYFINDER_AMD_LIB=$REPO/rayfinder_amd/$lib python tools/$TOOL $SPP "$@" 2>&1 | grep -v "amdgpu.ids\|RAYFINDER_AMD_LIB"
  done
done
