#!/bin/bash
OUT=gpurun_out/r06_soak; mkdir -p $OUT
export PYTHONPATH=$PWD
python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu_after_anyhit_stack.log 2>&1; grep -E "passed|failed" $OUT/pytest_gpu_after_anyhit_stack.log | tail -1
timeout 1500 python tools/gpu_fuzz.py 700000 30000 > $OUT/fuzz30000_anyhit_stack.log 2>&1; tail -1 $OUT/fuzz30000_anyhit_stack.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_try3.json 2> gpurun_out/r06_bench_try3.err; python - <<'PY'
import json
j = json.loads(open("gpurun_out/r06_bench_try3.json").read().strip().splitlines()[-1])
print(j["value"], j["repeats"]["value"], j["kernel_ms_rank0"], j.get("parity_crop", {}).get("verdict"), {k: v.get("value") for k, v in j["regimes"].items() if isinstance(v, dict)})
PY
