#!/bin/bash
# usage (anywhere, hipcc cross-compiles): tools/build_variant.sh <name> "<EXTRA defines>"  ->  rtiow-rust_amd/csrc/variants/<name>.so
# A/B runs on the GPU box then pick a build with RTIOW_GPU_LIB (tools/ab.sh): no GPU-minutes spent compiling.
set -e
cd "$(dirname "$0")/../rtiow-rust_amd/csrc"
mkdir -p variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function $2 -shared -o variants/$1.so rtg_api.hip scene_builder.cpp
echo "built variants/$1.so ($2)"
