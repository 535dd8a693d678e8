#!/bin/bash
OUT=gpurun_out/r06_lanes; mkdir -p $OUT
export PYTHONPATH=$PWD
python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu_full2.log 2>&1; grep -E "passed|failed" $OUT/pytest_gpu_full2.log | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_try2.json 2> gpurun_out/r06_bench_try2.err; grep "^\[bench\]" gpurun_out/r06_bench_try2.err | tail -12; python - <<'PY'
import json
j = json.loads(open("gpurun_out/r06_bench_try2.json").read().strip().splitlines()[-1])
print(j["value"], j["repeats"]["value"], j["kernel_ms_rank0"], j.get("parity_crop", {}).get("verdict"))
PY
