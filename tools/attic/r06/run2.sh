#!/bin/bash
# round 6, GPU call 4: (a) new parity tests, (b) refill thresholds on the three stand-ins, base library against the round-6 kernel, (c) f32 transcendentals timing,
# (d) per-bounce counters of three refill thresholds, (e) ray order through an index list on the x8 scene
OUT=gpurun_out/r06_lanes; mkdir -p $OUT gpurun_out/r06_f32 gpurun_out/r06_sort_x8
export PYTHONPATH=$PWD
RF_F32_REPORT=$PWD/gpurun_out/r06_f32/grades.json python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "inv_stream or f32_transcendentals or local_transport" > $OUT/pytest_new.log 2>&1; tail -3 $OUT/pytest_new.log
echo "== plain: bounce-1 / shadow refill threshold (round 6 kernel)"
python tools/r06/ab_variants.py 64 "refill_min=40,refill_min_deep=12" "refill_min=32,refill_min_deep=12" "refill_min=24,refill_min_deep=12" "refill_min=16,refill_min_deep=12" "refill_min=48,refill_min_deep=12" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_refill_b1.log
echo "== f32 transcendentals (plain, 64 spp)"
python tools/r06/ab_variants.py 64 "transcendentals=0" "transcendentals=1" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_f32/ab_f32_plain.log
for sc in clutter x8; do
  if [ $sc = x8 ]; then export RF_SCENE_SCALE=8 RF_SCENE_DETAIL=plain; SPP=16; else export RF_SCENE_SCALE=1 RF_SCENE_DETAIL=clutter; SPP=64; fi
  echo "== $sc: base"
  RAYFINDER_AMD_LIB=$PWD/rayfinder_amd/librayfinder_amd_base.so python tools/r06/ab_variants.py $SPP "refill_min_deep=22" "refill_min_deep=12" 2>&1 | grep -v "amdgpu.ids\|RAYFINDER_AMD_LIB"
  echo "== $sc: round 6 kernel"
  python tools/r06/ab_variants.py $SPP "refill_min_deep=22" "refill_min_deep=12" "refill_min_deep=6" 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $OUT/ab_refill_other_scenes.log
unset RF_SCENE_SCALE RF_SCENE_DETAIL
echo "== per-bounce counters, three thresholds"
bash tools/pmc_bounces.sh $PWD/$OUT/refill_counters.csv 64 "refill_min_deep=22" "refill_min_deep=12" "refill_min_deep=1" > $OUT/refill_counters.txt 2>&1; tail -50 $OUT/refill_counters.txt
echo "== x8: order through an index list"
RF_SORT_INDIRECT=1 python tools/gpu_sort_potential.py 128 64 plain 8 > gpurun_out/r06_sort_x8/potential_x8_indirect.log 2>&1; tail -30 gpurun_out/r06_sort_x8/potential_x8_indirect.log
