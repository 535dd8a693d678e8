#!/bin/bash
OUT=gpurun_out/r06_raygen; mkdir -p $OUT
export PYTHONPATH=$PWD
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not bench_line and not config5" > $OUT/pytest_fastdiv.log 2>&1; grep -E "passed|failed" $OUT/pytest_fastdiv.log | tail -1
python tools/r06/ab_variants.py 320 "dense_raygen=0" "dense_raygen=1" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_fastdiv_320spp.log
timeout 600 python tools/gpu_fuzz.py 400000 3000 > $OUT/fuzz3000_fastdiv.log 2>&1; tail -1 $OUT/fuzz3000_fastdiv.log
