#!/bin/bash
# A/B of the default library against an experiment build (make lib EXP=... LIBNAME=librayfinder_amd_exp.so), interleaved in one gpurun call:
#   bash tools/ab_libs.sh <spp> "<variant>" ["<variant>" ...]      (variants as for tools/gpu_opt2.py)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
SPP=$1; shift
for rep in 1 2; do
  for lib in librayfinder_amd.so librayfinder_amd_exp.so; do
    echo "== $lib (pass $rep)"
    RAYFINDER_AMD_LIB=$REPO/rayfinder_amd/$lib python tools/gpu_opt2.py $SPP "$@" 2>&1 | grep "Mrays\|MISMATCH"
  done
done
