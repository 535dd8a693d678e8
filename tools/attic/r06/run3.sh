#!/bin/bash
# round 6, GPU call 5: the whole GPU suite on the round-6 kernel, a fuzz soak, the index-list ray order re-measured (timing span fixed)
OUT=gpurun_out/r06_lanes; mkdir -p $OUT gpurun_out/r06_sort_x8
export PYTHONPATH=$PWD
python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu_full.log 2>&1; tail -4 $OUT/pytest_gpu_full.log
RF_SORT_INDIRECT=1 python tools/gpu_sort_potential.py 128 64 plain 8 > gpurun_out/r06_sort_x8/potential_x8_indirect.log 2>&1; tail -24 gpurun_out/r06_sort_x8/potential_x8_indirect.log
timeout 1500 python tools/gpu_fuzz.py 100000 8000 > $OUT/fuzz8000.log 2>&1; tail -3 $OUT/fuzz8000.log
