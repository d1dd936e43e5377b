#!/bin/bash
# Time budget of the path kernel by ablation / doubling (timing only: most variants render wrong images).
#   git apply tools/ablation/time_budget.patch      # adds the ABL_* switches to csrc/ssx_kernels.hip -- made against round 3's kernel (commit ad8295a: apply it in a
#                                                   # worktree of that commit; it does not apply to the round-4 source, whose fold and hand-over were rewritten)
#   tools/ablation/run.sh build                     # here: one library per switch
#   gpurun -- 'tools/ablation/run.sh bench'         # on the GPU box: product and every variant, twice, on one box
# A variant is only a measurement if it leaves the path structure alone (same rays, same shading decisions): removing code
# whose results decide nothing (shadow traces: visibility only scales radiance; the fold; the level stores) or running a
# region twice with the first result kept alive.  Two traps met on the way (profiles/r03_end/time_budget.log):
#   * removing the stores of a result lets the compiler remove what computed it (no fs/np/vis stores -> no shadow trace);
#   * a textured quad given a constant zero albedo ends its paths early (ABL_NOTEX; ABL_NOTEX2 keeps a non-zero table).
export SSX_DEBUG_ENV=1 # the master switch of the A/B environment variables (README)
VARIANTS="acos:-DABL_ACOS acossc:-DABL_ACOS,-DABL_SINCOS noshadow:-DABL_NOSHADOW nofold:-DABL_NOFOLD notex2:-DABL_NOTEX2 trace2x:-DABL_TRACE2X light2x:-DABL_LIGHT2X bsdf2x:-DABL_BSDF2X albedo2x:-DABL_ALBEDO2X nee2x:-DABL_NEE2X smalllog:-DABL_SMALLLOG nt:-DABL_NT aos:-DABL_LOG_AOS"
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R
if [ "$1" = build ]; then
  for v in $VARIANTS; do tools/build_variant.sh abl_${v%%:*} $(echo ${v#*:} | tr , ' ') | tail -1; done
  exit
fi
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["stage_ms"])'
for round in 1 2; do
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$P" product
  for v in $VARIANTS; do n=${v%%:*}
    SSX_HIP_LIB_OVERRIDE=$R/simple_spectral_amd/libssx_hip_abl_$n.so python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$P" $n
  done
done
