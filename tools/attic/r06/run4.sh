#!/bin/bash
OUT=gpurun_out/r06_texel; mkdir -p $OUT
export PYTHONPATH=$PWD
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "texel_tiles or render_path_traversal or bench_line_shape or random_scenes" > $OUT/pytest_texel.log 2>&1; tail -3 $OUT/pytest_texel.log
bash tools/r06/texel_ab.sh
RF_FUZZ_TILES=1 timeout 600 python tools/gpu_fuzz.py 200000 2000 > $OUT/fuzz2000_tiles.log 2>&1; tail -2 $OUT/fuzz2000_tiles.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_try1.json 2> gpurun_out/r06_bench_try1.err; tail -c 3000 gpurun_out/r06_bench_try1.err; python - <<'PY'
import json
j = json.loads(open("gpurun_out/r06_bench_try1.json").read().strip().splitlines()[-1])
print(j["value"], j["repeats"]["value"], j["kernel_ms_rank0"], j.get("parity_crop", {}).get("verdict"))
print("traced only", j.get("value_traced_only"), "shortcuts off", j.get("shortcuts_off"), "\nf32", j.get("f32_transcendentals"), "\ncurve", j.get("batch_depth_curve"))
print("roofline bound", j["roofline"].get("bound"), j["roofline"].get("frac"), j["roofline"].get("frac_of_binding_ceiling"), j["roofline"].get("bound_by_phase"))
print("device_memory", j["device_memory"]); print({k: (v.get("value") if isinstance(v, dict) else None) for k, v in j.get("regimes", {}).items()})
PY
