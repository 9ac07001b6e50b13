#!/bin/bash
# Build kernel variants side by side: tools/variants.sh <name> "-DFOO=1 ..." -> variants/libksa_<name>.so
# Variant builds carry -DKSA_EXPERIMENTS: only they read the KSA_* environment switches (the product library reads none).
# Run one with tools/with_lib.sh variants/libksa_<name>.so <command>.
set -e
cd "$(dirname "$0")/.."
mkdir -p variants
name=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-unused-value -shared -fPIC -DKSA_EXPERIMENTS $@ -o variants/libksa_$name.so prgs-sdr-kspecanal_amd/csrc/ksa_api.hip
echo built variants/libksa_$name.so
