#!/bin/bash
# VGPR / SGPR / LDS / scratch of every kernel instantiation (device-only compile of rf_trace.hip and rf_shade.hip, same flags as the Makefile).
# usage: tools/kernel_resources.sh [extra -D flags]        (the last unit's code object stays at $TMPDIR/rf_renderer.dev.o.co for llvm-objdump)
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/rf_renderer.dev.o
for UNIT in rf_shade rf_trace; do
/opt/rocm/bin/hipcc -std=c++20 -O3 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-gpu-flush-denormals-to-zero -I$REPO/include "$@" --offload-device-only -c $REPO/rayfinder_amd/csrc/$UNIT.hip -o $OUT || exit 1
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$OUT --output=$OUT.co || exit 1
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $OUT.co | python3 -c "
import sys, re
txt = sys.stdin.read()
for b in txt.split('- .agpr_count')[1:]:
    name = re.search(r'\.name:\s+(\S+)', b).group(1)
    g = lambda k: re.search(r'\.' + k + r':\s+(\d+)', b).group(1)
    import subprocess
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().replace('rf::(anonymous namespace)::', '').split('(')[0]
    print(f'{dem[:60]:60s} vgpr {g(\"vgpr_count\"):>4s} agpr {b.split()[0].strip(\":\"):>3s} sgpr {g(\"sgpr_count\"):>4s} lds {g(\"group_segment_fixed_size\"):>6s} scratch {g(\"private_segment_fixed_size\"):>4s}')
" | sort
done
