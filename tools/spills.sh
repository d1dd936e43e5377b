#!/bin/bash
# where are the VGPR spills of a kernel? usage: spills.sh kernel_name [flags]
K=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -S --cuda-device-only -gline-tables-only -DSSX_PROBE_BUILD "$@" -o /tmp/kg.s /root/repo/simple_spectral_amd/csrc/ssx_kernels.hip 2>/dev/null
python3 - $K <<'PY'
import re,sys
t=open('/tmp/kg.s').read()
files={}
for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?',t):
    files[int(m.group(1))]=(m.group(3) or m.group(2))
m=re.search(r"^%s:[^\n]*\n(.*?)\n\.Lfunc_end"%sys.argv[1],t,re.S|re.M)
loc=None
for l in m.group(1).split("\n"):
    mm=re.match(r"\s*\.loc\s+(\d+)\s+(\d+)",l)
    if mm: loc=(files.get(int(mm.group(1)),'?').split('/')[-1],int(mm.group(2)))
    if 'scratch_' in l: print(loc,l.strip())
PY
