#!/bin/bash
# round 4, GPU call 1: parity of the non-blocking pixel sums, rank shares before / after, A/B against round 3's library
O=gpurun_out/r04_a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
SSX_HIP_LIB_OVERRIDE=$PWD/simple_spectral_amd/libssx_hip_r03.so python tools/rank_share.py --configs headline --tag r03-spin > $O/rank_share_r03.log 2>&1
python tools/rank_share.py --tag r04-chain > $O/rank_share_r04.log 2>&1
cat $O/rank_share_r03.log $O/rank_share_r04.log | grep '^#'
tools/ab_bench.sh simple_spectral_amd/libssx_hip_r03.so > $O/ab_r03_vs_r04.log 2>&1; cat $O/ab_r03_vs_r04.log | cut -c1-60
SSX_HIP_LIB_OVERRIDE=$PWD/simple_spectral_amd/libssx_hip_formal.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "pixel_sums or config1 or many_units or bit_exact_against" > $O/pytest_formal.log 2>&1; tail -2 $O/pytest_formal.log
