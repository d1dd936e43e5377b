#!/bin/bash
# Builds libssx_hip.so of a git revision into simple_spectral_amd/libssx_hip_<name>.so (for tools/ab_bench.sh).
# usage: tools/build_rev.sh <rev> <name>
REV=${1:-HEAD}; NAME=${2:-prev}
D=$(mktemp -d)
mkdir -p $D/simple_spectral_amd/csrc $D/include
for f in $(git ls-tree -r --name-only $REV simple_spectral_amd/csrc include); do git show $REV:$f > $D/$f; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared $D/simple_spectral_amd/csrc/ssx_api.hip -o simple_spectral_amd/libssx_hip_$NAME.so -lpthread && echo built simple_spectral_amd/libssx_hip_$NAME.so
rm -rf $D
