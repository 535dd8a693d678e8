#!/bin/bash
# Round 5's profile set, from the build in the tree (one gpurun call, ~25 GPU-minutes):
#   gpurun_out/r05_final  = tools/final_profile.sh (tests, calibrations, counters of the driver command, bench lines of configs 2 / 3 / 5 -- the driver command's line now
#                           carries the `regimes` block and the cold start --, shard emulation WITH the priced exchange) + the clutter stand-in's own bench line
#                           + the self-launched / RCCL-loop-back bench line (exchange_ms per repeat)
#   gpurun_out/r05_hbm2   = the same counter recipe on the out-of-cache atrium (--scene-scale 8), calibration reused
# Afterwards: bash tools/copy_final_profile_r05.sh (container side).
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
bash tools/final_profile.sh gpurun_out/r05_final
F=$REPO/gpurun_out/r05_final
python bench.py --steps 20 --warmup 5 --scene-detail clutter > $F/bench_clutter.json 2> /dev/null
python bench.py --gpus 1 --launch --exchange-at-world-1 --steps 20 --warmup 5 --no-cpu-baseline > $F/bench_self_launched_rccl_loopback.json 2> $F/bench_self_launched.err
STEPS=4 WARMUP=1 BENCH_ARGS="--scene-scale 8" CALIB_FROM=gpurun_out/r05_final/roofline bash tools/roofline_pmc.sh $REPO/gpurun_out/r05_hbm2 > $REPO/gpurun_out/r05_hbm2.log 2>&1
cp $REPO/gpurun_out/r05_hbm2/pmc_per_ray.json $REPO/profiles/pmc_per_ray_x8.json
python bench.py --scene-scale 8 --steps 4 --warmup 1 --cpu-seconds 6 > $REPO/gpurun_out/r05_hbm2/bench.json 2> $REPO/gpurun_out/r05_hbm2/bench.err
cp $REPO/profiles/pmc_per_ray.json $REPO/profiles/pmc_per_ray_x8.json $F/
for f in clutter x8 bench_driver_command loopback; do python - $f <<'PY'
import json, sys
p = {"clutter": "gpurun_out/r05_final/bench_clutter.json", "x8": "gpurun_out/r05_hbm2/bench.json", "bench_driver_command": "gpurun_out/r05_final/bench_driver_command.json",
     "loopback": "gpurun_out/r05_final/bench_self_launched_rccl_loopback.json"}[sys.argv[1]]
try:
    j = json.loads(open(p).read().strip().splitlines()[-1])
    a = j["roofline"].get("algorithmic", {})
    print(sys.argv[1], j["value"], j["repeats"]["value"], j.get("parity_crop", {}).get("verdict"), "bound", j["roofline"].get("bound"), j["roofline"].get("frac"),
          "valu", j["roofline"].get("ceilings", {}).get("valu", {}).get("frac"), "visits", a.get("node_visits_per_ray"), "tri tests", a.get("triangle_tests_per_ray"), j["kernel_ms_rank0"],
          "exchange_ms", j.get("exchange_ms"), "regimes", {k: (v.get("value"), v.get("parity_crop", {}).get("verdict")) for k, v in (j.get("regimes") or {}).items() if isinstance(v, dict)},
          "cold", (j.get("occluder_cache") or {}).get("value_cold_start"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
timeout 900 python tools/gpu_fuzz.py 50000 20000 > $F/fuzz20000.log 2>&1; tail -1 $F/fuzz20000.log
