#!/bin/bash
# AddressSanitizer + UBSan over the host-side parsers (CPU build only; GPU ASan is not available on this pool):
#   the JPEG decoder on corrupted JPEGs, and the GLB / PNG / .pt ingest on corrupted Duck.glb / Duck.pt.
# Needs Pillow to make the corpus.  Usage: bash tools/sanitize/run.sh   (from the repo root, in the build container)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
W=${TMPDIR:-/tmp}/rf_sanitize; rm -rf $W; mkdir -p $W/jpeg $W/ingest
FLAGS="-std=c++20 -O1 -g -fwrapv -fsanitize=address,undefined -fno-sanitize-recover=undefined -I$ROOT/rayfinder_amd/csrc -I$ROOT/include -I/opt/rocm/include -D__HIP_PLATFORM_AMD__"
g++ $FLAGS $ROOT/tools/sanitize/jpeg_driver.cpp $ROOT/rayfinder_amd/csrc/rf_jpeg.cpp -o $W/jpeg_driver
g++ $FLAGS $ROOT/tools/sanitize/ingest_driver.cpp $ROOT/rayfinder_amd/csrc/rf_gltf.cpp $ROOT/rayfinder_amd/csrc/rf_pt_format.cpp \
    $ROOT/rayfinder_amd/csrc/rf_bvh.cpp $ROOT/rayfinder_amd/csrc/rf_jpeg.cpp -lz -o $W/ingest_driver
python3 $ROOT/tools/sanitize/make_corpus.py $ROOT $W
$W/jpeg_driver $W/jpeg/*
$W/ingest_driver $W/ingest/*
echo "sanitizers: clean"
