cd $GRAFT_REPO_ROOT
export SSX_DEBUG_ENV=1
for V in pfr pff pfb; do
  echo "== parity $V"; SSX_HIP_LIB_OVERRIDE=$PWD/simple_spectral_amd/libssx_hip_$V.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bit_exact_against_oracle or goldens or pixel_sums_chain or many_units" 2>&1 | tail -2
done
tools/ab_bench.sh simple_spectral_amd/libssx_hip_pfr.so simple_spectral_amd/libssx_hip_pff.so simple_spectral_amd/libssx_hip_pfb.so > gpurun_out/r05_ab_prefetch.log 2>&1; cat gpurun_out/r05_ab_prefetch.log
