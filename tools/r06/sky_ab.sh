#!/bin/bash
# kSky: misses per thread and trip (RF_EXP_SKY_UNROLL builds); ms_shade of the stats = kShade + kSky
export PYTHONPATH=$PWD
for rep in 1 2; do for l in librayfinder_amd.so librayfinder_amd_sky2.so librayfinder_amd_sky4.so; do echo "== $l"; RAYFINDER_AMD_LIB=$PWD/rayfinder_amd/$l python tools/r06/ab_variants.py 320 "-" 2>&1 | grep -v "amdgpu.ids\|RAYFINDER_AMD_LIB" | tail -1; done; done
cd /tmp && export TMPDIR=/tmp
for l in librayfinder_amd.so librayfinder_amd_sky2.so librayfinder_amd_sky4.so; do
  D=/tmp/skytr_$l; rm -rf $D
  RAYFINDER_AMD_LIB=$OLDPWD/rayfinder_amd/$l timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o r -- python $OLDPWD/tools/gpu_variant_bounces.py 64 "-" > /dev/null 2>&1
  f=$(find $D -name '*kernel_stats.csv' | head -1); echo "$l: $(grep kSky $f | cut -d, -f2-4)"
done
