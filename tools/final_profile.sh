#!/bin/bash
# Runs on the GPU box (one gpurun call): everything profiles/r02_final holds, from the build in the tree.
#   bash tools/final_profile.sh [outdir]      (default gpurun_out/final)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$REPO/gpurun_out/final}
case $OUT in /*) ;; *) OUT=$REPO/$OUT ;; esac
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -n 2 $OUT/pytest_gpu.log
# counters first: bench.py's roofline block reads profiles/pmc_per_ray.json, which must come from THIS build
STEPS=20 WARMUP=20 bash tools/roofline_pmc.sh $OUT/roofline > $OUT/roofline.log 2>&1
cp $OUT/roofline/pmc_per_ray.json $REPO/profiles/pmc_per_ray.json
cd $REPO
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err
python bench.py > $OUT/bench_default.json 2> /dev/null
python bench.py --scene tests/golden/Duck.glb --width 800 --height 600 --bounces 4 --steps 4 --warmup 1 > $OUT/bench_config2.json 2> /dev/null
python bench.py --width 3840 --height 2160 --bounces 16 --steps 16 --warmup 1 --cpu-seconds 6 > $OUT/bench_config5.json 2> /dev/null
timeout 600 python tools/gpu_shard_emulation.py 320 > $OUT/shard_emulation.log 2> /dev/null
for f in bench_driver_command bench_default bench_config2 bench_config5; do python - $OUT/$f.json <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], j["value"], j.get("parity_crop", {}).get("verdict"), j["roofline"].get("frac"), j["roofline"].get("l1", {}).get("frac"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
cat $OUT/shard_emulation.log
