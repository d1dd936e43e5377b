#!/bin/bash
# Round 6, ninth GPU call: the fold loading the next pass's tail words a pass ahead (libssx_hip_tailpre.so, built in the build container) against the product
O=gpurun_out/r06; mkdir -p $O
export SSX_DEBUG_ENV=1
BENCH_ARGS="--scene plane-srgb --res 1024 --spp 1024 --scratch-cap-gb 20" bash tools/ab_bench.sh simple_spectral_amd/libssx_hip_tailpre.so 2>&1 | cut -c1-170
bash tools/ab_bench.sh simple_spectral_amd/libssx_hip_tailpre.so 2>&1 | cut -c1-170
SSX_HIP_LIB_OVERRIDE=$PWD/simple_spectral_amd/libssx_hip_tailpre.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py -m gpu -q -rf > $O/pytest_tailpre.log 2>&1; echo "pytest tailpre rc=$?"; grep -E "^FAILED|^ERROR| passed| failed" $O/pytest_tailpre.log | cut -c1-300
