#!/bin/bash
# A/B on ONE GPU box (box-to-box spread is +-3 %): alternate bench.py runs of the product library and of the
# variant libraries given as arguments (paths, loaded through SSX_HIP_LIB_OVERRIDE), three rounds each.
# usage: [BENCH_ARGS="--scene plane-srgb ..."] tools/ab_bench.sh [variant.so ...]   -> prints value / ms_per_step / kernel_ms per run
export SSX_DEBUG_ENV=1 # the master switch of the A/B environment variables (README)
R=$(pwd)
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["stage_ms"], d["roofline"]["scratch_bytes"])'
for round in 1 2 3; do
	python $R/bench.py --steps 10 --warmup 2 --quick $BENCH_ARGS 2>/dev/null | python -c "$P" product
	for V in "$@"; do
		SSX_HIP_LIB_OVERRIDE=$R/$V python $R/bench.py --steps 10 --warmup 2 --quick $BENCH_ARGS 2>/dev/null | python -c "$P" $V
	done
done
