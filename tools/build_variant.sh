#!/bin/bash
# Builds the working tree's libssx_hip.so with extra compiler flags into simple_spectral_amd/libssx_hip_<name>.so (for tools/ab_bench.sh).
# usage: tools/build_variant.sh <name> [-DMACRO ...]
NAME=$1; shift
python -m simple_spectral_amd.build --embed-only
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared "$@" simple_spectral_amd/csrc/ssx_api.hip -o simple_spectral_amd/libssx_hip_$NAME.so -lpthread -ldl && echo built simple_spectral_amd/libssx_hip_$NAME.so
