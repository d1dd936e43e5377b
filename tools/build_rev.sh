#!/bin/bash
# Builds libssx_hip.so of a git revision into simple_spectral_amd/libssx_hip_<name>.so (for tools/ab_bench.sh).
# usage: tools/build_rev.sh <rev> <name>
REV=${1:-HEAD}; NAME=${2:-prev}
D=$(mktemp -d)
mkdir -p $D/simple_spectral_amd/csrc $D/include
for f in $(git ls-tree -r --name-only $REV simple_spectral_amd/csrc include); do git show $REV:$f > $D/$f; done
# revisions with the run-time specialisation embed their kernel sources (simple_spectral_amd/build.py: embed_sources)
if [ -f $D/simple_spectral_amd/csrc/ssx_jit.h ]; then
python3 - $D <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from simple_spectral_amd import build as b
d = sys.argv[1]
b.EMBED = tuple((n, p.replace(b.ROOT, d)) for n, p in b.EMBED)
b.SOURCES_GEN = b.SOURCES_GEN.replace(b.ROOT, d)
b.ROOT = d
b.embed_sources()
PY
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared $D/simple_spectral_amd/csrc/ssx_api.hip -o simple_spectral_amd/libssx_hip_$NAME.so -lpthread -ldl && echo built simple_spectral_amd/libssx_hip_$NAME.so
rm -rf $D
