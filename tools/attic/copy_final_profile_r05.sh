#!/bin/bash
# After `gpurun -- bash tools/final_profile_r05.sh`: copy what was merged into gpurun_out/ into the tracked profiles/ directories (container side).
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd $REPO
F=gpurun_out/r05_final; P=profiles/r05_final
mkdir -p $P profiles/r05_hbm/x8_counters
cp $F/roofline/*.txt $F/roofline/*.csv $F/roofline/*.json $F/roofline/*.jsonl $F/roofline/trace.log $P/
cp $F/bench_*.json $F/shard_emulation.log $F/fuzz20000.log $P/
grep -h "passed\|failed\|error" $F/pytest_gpu.log | tail -3 > $P/pytest_gpu.txt
cp $F/pmc_per_ray.json profiles/pmc_per_ray.json
cp $F/pmc_per_ray_x8.json profiles/pmc_per_ray_x8.json
for f in $(ls gpurun_out/r05_hbm2 | grep -v bench.err); do cp gpurun_out/r05_hbm2/$f profiles/r05_hbm/x8_counters/$f; done
sed -i 's/"profile": "roofline"/"profile": "r05_final"/' profiles/pmc_per_ray.json $P/pmc_per_ray.json $P/per_bounce.json
sed -i 's/"profile": "r05_hbm2"/"profile": "r05_hbm"/' profiles/pmc_per_ray_x8.json profiles/r05_hbm/x8_counters/pmc_per_ray.json profiles/r05_hbm/x8_counters/per_bounce.json 2>/dev/null || true
git status --short profiles | wc -l
