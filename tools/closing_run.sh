#!/bin/bash
# closing run of a round on the GPU box (tools/closing_run.sh <tag>): whole GPU suite, bench line, rocprofv3 stats + counters, lane statistics, region times, run-time specialisation,
# the GPU half of the sanitizer pass
O=gpurun_out/${1:-closing}; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E " passed| failed" $O/pytest_gpu.log | tail -1
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print({k:d[k] for k in ('value','ms_per_step','value_device_resident')}, d['roofline']['frac'], d['roofline']['stage_ms'], d['cpu_baseline']['value'])"
bash tools/profile_round.sh ${1:-closing}_prof > $O/profile_round.log 2>&1; tail -2 $O/profile_round.log | cut -c1-200
python tools/lanestat.py > $O/lanestat.log 2>&1; tail -22 $O/lanestat.log
python tools/jit_rate.py > $O/jit_rate.log 2>&1; tail -3 $O/jit_rate.log | cut -c1-300
python tools/regtime.py --census profiles/r05/isa_census.csv > $O/regtime.log 2>&1; tail -13 $O/regtime.log
bash tools/sanitize.sh --gpu-only $O/sanitize_gpu.log > /dev/null 2>&1; echo "sanitize rc=$?"; tail -12 $O/sanitize_gpu.log
bash tools/test_kernel_variants.sh > $O/parity_per_kernel_variant.log 2>&1; cat $O/parity_per_kernel_variant.log
python tools/fuzz_scenes.py 100000 4000 > $O/fuzz_scenes.log 2>&1; tail -2 $O/fuzz_scenes.log
python tools/bench_sweep.py "--res 128 --spp 16" "--scene cornell --spp 1024 --uplift jh" "--scene plane-srgb --res 1024 --spp 1024" "--res 2048 --spp 2048 --observer 2006 --steps 2 --warmup 1" "--texture procedural:4096" "--observer 2006" > $O/configs.log 2>&1; cut -c1-130 $O/configs.log
