#!/bin/bash
# round 6 fuzz soak on the final build: N seeds of the randomized differential test (GPU image == oracle image bit for bit)
OUT=gpurun_out/r06_soak; mkdir -p $OUT
export PYTHONPATH=$PWD
timeout ${2:-3000} python tools/gpu_fuzz.py ${3:-500000} ${1:-100000} > $OUT/fuzz_soak.log 2>&1; tail -2 $OUT/fuzz_soak.log
