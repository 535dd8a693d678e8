#!/bin/bash
OUT=gpurun_out/r06_lanes; mkdir -p $OUT
export PYTHONPATH=$PWD
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "any_hit or occluder or render_path_traversal or random_scenes or shadow or atrium_crops" > $OUT/pytest_anyhit_stack.log 2>&1; grep -E "passed|failed" $OUT/pytest_anyhit_stack.log | tail -1
echo "== any-hit kernels: grid of 6 per CU (persistent_blocks forced) against each kernel's own residency"
python tools/r06/ab_variants.py 64 "persistent_blocks=1536" "persistent_blocks=0" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_anyhit_grid.log
for lib in librayfinder_amd.so librayfinder_amd_ah7.so librayfinder_amd_lr4.so librayfinder_amd_lr6.so librayfinder_amd_lr12.so librayfinder_amd.so; do
  echo "== $lib"
  RAYFINDER_AMD_LIB=$PWD/rayfinder_amd/$lib python tools/r06/ab_variants.py 64 "-" 2>&1 | grep -v "amdgpu.ids\|RAYFINDER_AMD_LIB" | tail -1
done 2>&1 | tee $OUT/ab_leaf_repeat_libs.log
RF_SCENE_DETAIL=clutter python tools/r06/ab_variants.py 64 "persistent_blocks=1536" "persistent_blocks=0" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_anyhit_grid_clutter.log
