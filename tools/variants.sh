#!/bin/bash
# Build kernel variants side by side: tools_variants.sh name "-DFOO=1 ..." -> gpurun_out/../variants/libksa_<name>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p variants
name=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-unused-value -shared -fPIC $@ -o variants/libksa_$name.so prgs-sdr-kspecanal_amd/csrc/ksa_api.hip
echo built variants/libksa_$name.so
