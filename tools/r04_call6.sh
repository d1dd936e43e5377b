#!/bin/bash
O=gpurun_out/r04_f; mkdir -p $O
export SSX_DEBUG_ENV=1
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log | cut -c1-300
echo "== formal"; SSX_HIP_LIB_OVERRIDE=$PWD/simple_spectral_amd/libssx_hip_formal.so python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "pixel_sums or config1 or many_units or bit_exact_against or launch_chunking" 2>&1 | tail -1 | cut -c1-200
tools/ab_bench.sh simple_spectral_amd/libssx_hip_r03.so > $O/ab.log 2>&1; cut -c1-110 $O/ab.log
python tools/rank_share.py --tag r04-final > $O/rank_share.log 2>&1; grep "^#" $O/rank_share.log
