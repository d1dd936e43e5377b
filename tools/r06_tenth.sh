#!/bin/bash
# Round 6, tenth GPU call: this round's library against round 5's final one (343ac83, built in the build container) on the headline workload, one box
export SSX_DEBUG_ENV=1
bash tools/ab_bench.sh simple_spectral_amd/libssx_hip_r05.so 2>&1 | cut -c1-170
bash tools/pmc_quick.sh "" simple_spectral_amd/libssx_hip_r05.so 2>&1 | grep "render_kernel_cornell" | grep "INSTS_VALU\|WAVE_CYCLES\|INSTS_SALU\|INSTS_LDS"
