# round 5: kShade with the 64-byte attribute record + hit point from the closest-hit launch (librayfinder_amd.so) against the 128-byte record (librayfinder_amd_base.so):
# per-kernel counters of one 32-spp frame of the plain atrium (averages per dispatch)
mkdir -p gpurun_out/r05_shade
for lib in librayfinder_amd_base.so librayfinder_amd.so; do
  echo "==== $lib"
  RAYFINDER_AMD_LIB=$PWD/rayfinder_amd/$lib bash tools/pmc_pass.sh "python $PWD/tools/gpu_variant_bounces.py 32 -" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" 2>&1 | grep -E "^##|kShade|kTraceWide<false|kSky"
done > gpurun_out/r05_shade/record64_counters.txt 2>&1
cat gpurun_out/r05_shade/record64_counters.txt
