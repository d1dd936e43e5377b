#!/bin/bash
# Round 6, eleventh GPU call: the conservative pass 1 (contracted shear / edge forms, sign test widened by the bound) against the exact one
# (libssx_hip_p1exact.so = the previous commit's library): parity suites, both workloads' A/B, instruction counters
O=gpurun_out/r06e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py -q -m gpu -rf -x > $O/pytest_parity.log 2>&1; echo "pytest rc=$?"; grep -E "^FAILED|^ERROR|^E  | passed| failed" $O/pytest_parity.log | tail -8 | cut -c1-300
bash tools/ab_bench.sh simple_spectral_amd/libssx_hip_p1exact.so 2>&1 | cut -c1-170 | tee $O/ab_cornell.log
BENCH_ARGS="--scene plane-srgb --res 1024 --spp 1024" bash tools/ab_bench.sh simple_spectral_amd/libssx_hip_p1exact.so 2>&1 | cut -c1-170 | tee $O/ab_plane.log
bash tools/pmc_quick.sh "" simple_spectral_amd/libssx_hip_p1exact.so 2>&1 | grep "render_kernel" | tee $O/pmc_cornell.log
python tools/lanestat.py > $O/lanestat.log 2>&1; grep -i "pass-2\|pass 2\|trips\|candidates" $O/lanestat.log | head -12
