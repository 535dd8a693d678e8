#!/bin/bash
# Round 3's profile set, from the build in the tree (one gpurun call, ~15 GPU-minutes):
#   gpurun_out/r03_final  = tools/final_profile.sh (tests, counters of the driver command, bench lines of configs 2 / 3 / 5, shard emulation)
#   gpurun_out/r03_hbm    = the same counter recipe on the out-of-cache atrium (--scene-scale 8: 17 M triangles), calibration reused
# Afterwards copy both directories to profiles/ (and pmc_per_ray.json / pmc_per_ray_x8.json, which bench.py reads).
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
bash tools/final_profile.sh gpurun_out/r03_final
STEPS=4 WARMUP=1 BENCH_ARGS="--scene-scale 8" CALIB_FROM=gpurun_out/r03_final/roofline bash tools/roofline_pmc.sh $REPO/gpurun_out/r03_hbm > $REPO/gpurun_out/r03_hbm.log 2>&1
cp $REPO/gpurun_out/r03_hbm/pmc_per_ray.json $REPO/profiles/pmc_per_ray_x8.json
python bench.py --scene-scale 8 --steps 4 --warmup 1 --cpu-seconds 6 > $REPO/gpurun_out/r03_hbm/bench.json 2> $REPO/gpurun_out/r03_hbm/bench.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r03_hbm/bench.json").read().strip().splitlines()[-1])
print("x8", j["value"], j.get("parity_crop", {}).get("verdict"), j["roofline"].get("bound"), j["roofline"].get("frac"))
PY
cp $REPO/profiles/pmc_per_ray.json $REPO/profiles/pmc_per_ray_x8.json $REPO/gpurun_out/r03_final/
