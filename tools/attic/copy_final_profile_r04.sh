#!/bin/bash
# After `gpurun -- bash tools/final_profile_r04.sh`: copy what was merged into gpurun_out/ into the tracked profiles/ directories
# (container side; the recipe's directory is called `roofline`, the profile name inside the json files is set to the directory's).
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd $REPO
F=gpurun_out/r04_final; P=profiles/r04_final
mkdir -p $P profiles/r04_hbm profiles/r04_lanes profiles/r04_valu
cp $F/roofline/*.txt $F/roofline/*.csv $F/roofline/*.json $F/roofline/*.jsonl $F/roofline/trace.log $P/
cp $F/bench_*.json $F/shard_emulation.log $F/fuzz20000.log $P/
grep -h "passed\|failed\|error" $F/pytest_gpu.log | tail -3 > $P/pytest_gpu.txt
cp $F/pmc_per_ray.json profiles/pmc_per_ray.json
cp $F/pmc_per_ray_x8.json profiles/pmc_per_ray_x8.json
for f in $(ls gpurun_out/r04_hbm | grep -v bench.err); do cp gpurun_out/r04_hbm/$f profiles/r04_hbm/$f; done
cp gpurun_out/r04_lanes/* profiles/r04_lanes/
cp $F/roofline/valu_calib_w*.jsonl profiles/r04_valu/
sed -i 's/"profile": "roofline"/"profile": "r04_final"/' profiles/pmc_per_ray.json $P/pmc_per_ray.json $P/per_bounce.json
git status --short profiles | wc -l
