cd $GRAFT_REPO_ROOT
export SSX_DEBUG_ENV=1
echo "== parity pfr2"; SSX_HIP_LIB_OVERRIDE=$PWD/simple_spectral_amd/libssx_hip_pfr2.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bit_exact_against_oracle or goldens or pixel_sums_chain or many_units" 2>&1 | tail -2
tools/ab_bench.sh simple_spectral_amd/libssx_hip_pfr2.so 2>&1 | tee gpurun_out/r05_ab_prefetch2.log
