#!/bin/bash
O=gpurun_out/r04_b; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
python tools/rank_share.py --tag r04-chain2 > $O/rank_share_r04.log 2>&1
grep '^#' $O/rank_share_r04.log; grep -o '"N": [0-9], "rank": [0-9].*' $O/rank_share_r04.log | cut -c1-200
tools/ab_bench.sh simple_spectral_amd/libssx_hip_r03.so > $O/ab_r03_vs_r04.log 2>&1; cat $O/ab_r03_vs_r04.log | cut -c1-60
SSX_HIP_LIB_OVERRIDE=$PWD/simple_spectral_amd/libssx_hip_formal.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "pixel_sums or config1 or many_units or bit_exact_against" > $O/pytest_formal.log 2>&1; tail -2 $O/pytest_formal.log
