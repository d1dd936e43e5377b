#!/bin/bash
# A/B on ONE GPU box between the product library as is and with an environment switch set (three rounds).
# usage: tools/ab_env.sh VAR=value [bench.py args...]
R=$(pwd); SW=$1; shift
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["stage_ms"])'
for round in 1 2 3; do
	python $R/bench.py --steps 10 --warmup 2 --quick "$@" 2>/dev/null | python -c "$P" default
	env $SW python $R/bench.py --steps 10 --warmup 2 --quick "$@" 2>/dev/null | python -c "$P" $SW
done
