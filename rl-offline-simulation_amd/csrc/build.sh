#!/bin/bash
# Builds liboffsim_hip.so (HIP kernels + C ABI) for gfx950, in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -I../../include \
 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero \
 -Wall -Wno-unused-function -Wno-int-to-pointer-cast"
# OUT=variants/libX.so build.sh -DFLAG ...  builds a variant next to the product library (A/B timing, profiling builds)
OUT=${OUT:-liboffsim_hip.so}
mkdir -p "$(dirname "$OUT")"
$HIPCC $FLAGS "$@" -o "$OUT" offsim_hip.hip
