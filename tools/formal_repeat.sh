#!/bin/bash
# The parity suites on the -DSSX_ACCUM_FORMAL build (simple_spectral_amd/libssx_hip_formal.so: the pixel sums' hand-over stated in the
# HIP memory model), N times in a row on one box, keeping the ids and assertions of whatever fails (-rf) -- VERDICT r05 item 1: the one
# red of profiles/r05/parity_per_kernel_variant.log had no name.
# usage: tools/formal_repeat.sh [N=10] [extra pytest args]   -> one block per run, then "green runs: g of N"
N=${1:-10}; shift
export SSX_DEBUG_ENV=1 # the master switch of the A/B environment variables (README): SSX_HIP_LIB_OVERRIDE is honoured only under it
LIB=$PWD/simple_spectral_amd/libssx_hip_formal.so
[ -f $LIB ] || { echo "no $LIB (python -m simple_spectral_amd.build)"; exit 1; }
G=0
for i in $(seq 1 $N); do
	echo "== formal run $i of $N"
	SSX_HIP_LIB_OVERRIDE=$LIB timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py -q -m gpu -rf -p no:cacheprovider "$@" > /tmp/formal_run_$i.log 2>&1
	rc=$?
	grep -E "^(FAILED|ERROR)|^E  | passed| failed| error" /tmp/formal_run_$i.log | cut -c1-400
	if [ $rc -eq 0 ]; then G=$((G+1)); else echo "-- rc=$rc; full log of the failing run follows"; sed -n '/=== FAILURES ===/,$p' /tmp/formal_run_$i.log | cut -c1-300 | head -150; fi
done
echo "green runs: $G of $N"
