#!/bin/bash
O=gpurun_out/r04_e; mkdir -p $O
export SSX_DEBUG_ENV=1
K="pixel_sums or config1 or many_units or bit_exact_against or launch_chunking or config4 or config5"
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "$K" 2>&1 | tail -2 | cut -c1-200
for V in sync formal; do echo "== $V"; SSX_HIP_LIB_OVERRIDE=$PWD/simple_spectral_amd/libssx_hip_$V.so python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "$K" 2>&1 | tail -2 | cut -c1-200; done
tools/ab_bench.sh simple_spectral_amd/libssx_hip_sync.so simple_spectral_amd/libssx_hip_r03.so > $O/ab.log 2>&1; cut -c1-110 $O/ab.log
python tools/rank_share.py --configs headline --tag r04-deferred > $O/rank_share.log 2>&1; grep "^#" $O/rank_share.log; grep -o '"N": [0-9], "rank": [0-9].*' $O/rank_share.log | cut -c1-260
SSX_HIP_LIB_OVERRIDE=$PWD/simple_spectral_amd/libssx_hip_sync.so python tools/rank_share.py --configs headline --tag r04-sync > $O/rank_share_sync.log 2>&1; grep "^#" $O/rank_share_sync.log
