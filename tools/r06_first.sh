#!/bin/bash
# Round 6, first GPU call: (1) name the red of the formal build, (2) first profile of the plane kernel.
O=gpurun_out/r06; mkdir -p $O
bash tools/formal_repeat.sh 10 > $O/formal_repeat.log 2>&1; tail -3 $O/formal_repeat.log
export BENCH_ARGS="--scene plane-srgb --res 1024 --spp 1024"
python bench.py --steps 10 --warmup 2 $BENCH_ARGS > $O/plane_bench.json 2> $O/plane_bench.err; cut -c1-600 $O/plane_bench.json; tail -3 $O/plane_bench.err
bash tools/profile_round.sh r06/plane_prof > $O/plane_profile_round.log 2>&1; tail -4 $O/plane_profile_round.log | cut -c1-300
python tools/lanestat.py --scene plane-srgb --res 1024 --spp 64 > $O/plane_lanestat.log 2>&1; tail -24 $O/plane_lanestat.log
python tools/regtime.py --scene plane-srgb --res 1024 --spp 64 > $O/plane_regtime.log 2>&1; tail -14 $O/plane_regtime.log
python bench.py --steps 10 --warmup 2 --quick > $O/cornell_quick.json 2>/dev/null; cut -c1-300 $O/cornell_quick.json
