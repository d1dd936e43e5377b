#!/bin/bash
O=gpurun_out/r04_d; mkdir -p $O
export SSX_DEBUG_ENV=1
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "background" > $O/pytest_jit.log 2>&1; tail -2 $O/pytest_jit.log | cut -c1-300
SSX_HIP_LIB_OVERRIDE=$PWD/simple_spectral_amd/libssx_hip_park.so python -m pytest tests/test_gpu_parity.py -m gpu -q -k "pixel_sums or config1 or many_units or bit_exact_against" 2>&1 | tail -1
tools/ab_bench.sh simple_spectral_amd/libssx_hip_park.so simple_spectral_amd/libssx_hip_r03.so > $O/ab_park.log 2>&1; cut -c1-110 $O/ab_park.log
