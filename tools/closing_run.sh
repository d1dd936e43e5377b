#!/bin/bash
# closing run of a round on the GPU box (tools/closing_run.sh <tag>): whole GPU suite, the headline and the plane bench lines, rocprofv3 stats + counters of
# both workloads, lane statistics, region times, the formal build's suites ten times over, every kernel variant, predicted shares, the other configurations,
# scene fuzz (random scenes and the warped built-in ones), the GPU half of the sanitizer pass, the multi-GPU dry run
# (FORMAL_N, FUZZ_FIRST, FUZZ_N, WARPED_FIRST, WARPED_N from the environment)
O=gpurun_out/${1:-closing}; mkdir -p $O $O/plane
python -m pytest tests -m gpu -q -rf > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "^FAILED|^ERROR| passed| failed" $O/pytest_gpu.log | tail -3 | cut -c1-300
python bench.py --steps 20 --warmup 5 --update-traffic > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print({k:d[k] for k in ('value','ms_per_step','value_device_resident')}, d['roofline']['frac'], d['roofline']['stage_ms'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['cpu_baseline']['reference_equivalent']['value'])"
python bench.py --steps 10 --warmup 2 --scene plane-srgb --res 1024 --spp 1024 --update-traffic > $O/plane/bench.json 2> $O/plane/bench.err; python -c "
import json; d=json.load(open('$O/plane/bench.json')); r=d['roofline']; print('plane', {k:d[k] for k in ('value','ms_per_step')}, r['frac'], r['frac_without_eliminated_work'], r['stage_ms'], r['traffic'], r['hbm']['traffic_bytes_per_sample'], d['check']['differing_floats'], d['cpu_baseline']['value'])"
bash tools/profile_round.sh ${1:-closing}/prof > $O/profile_round.log 2>&1; tail -2 $O/profile_round.log | cut -c1-200
BENCH_ARGS="--scene plane-srgb --res 1024 --spp 1024" bash tools/profile_round.sh ${1:-closing}/plane/prof > $O/plane/profile_round.log 2>&1; tail -2 $O/plane/profile_round.log | cut -c1-200
python tools/lanestat.py > $O/lanestat.log 2>&1; tail -24 $O/lanestat.log
python tools/lanestat.py --scene plane-srgb --res 1024 --spp 64 > $O/plane/lanestat.log 2>&1
python tools/regtime.py --census profiles/r05/isa_census.csv > $O/regtime.log 2>&1; tail -13 $O/regtime.log
python tools/regtime.py --scene plane-srgb --res 1024 --spp 64 --census profiles/r06/plane/isa_census.csv > $O/plane/regtime.log 2>&1; tail -13 $O/plane/regtime.log
python tools/jit_rate.py > $O/jit_rate.log 2>&1; tail -3 $O/jit_rate.log | cut -c1-300
python bench.py --dist-dry-run > $O/dry_run.json 2>/dev/null; grep -c dry_run $O/dry_run.json
bash tools/formal_repeat.sh ${FORMAL_N:-10} > $O/formal_repeat.log 2>&1; grep -E "passed|failed|green" $O/formal_repeat.log | cut -c1-120
bash tools/test_kernel_variants.sh > $O/parity_per_kernel_variant.log 2>&1; cat $O/parity_per_kernel_variant.log
python tools/rank_share.py --all-ranks --configs headline,plane > $O/rank_share.log 2>&1; tail -12 $O/rank_share.log | cut -c1-200
python tools/bench_sweep.py "--res 128 --spp 16" "--scene cornell --spp 1024 --uplift jh" "--scene plane-srgb --res 1024 --spp 1024" "--res 2048 --spp 2048 --observer 2006 --steps 2 --warmup 1" "--texture procedural:4096" "--observer 2006" > $O/configs.log 2>&1; cut -c1-130 $O/configs.log
python tools/fuzz_scenes.py ${FUZZ_FIRST:-130000} ${FUZZ_N:-6000} > $O/fuzz_scenes.log 2>&1; tail -2 $O/fuzz_scenes.log | cut -c1-400
python tools/fuzz_scenes.py --warped ${WARPED_FIRST:-210000} ${WARPED_N:-3000} > $O/fuzz_warped.log 2>&1; tail -2 $O/fuzz_warped.log | cut -c1-400
bash tools/sanitize.sh --gpu-only $O/sanitize_gpu.log > /dev/null 2>&1; echo "sanitize rc=$?"; tail -6 $O/sanitize_gpu.log
python tools/stress_parity.py > $O/stress_parity.log 2>&1; tail -2 $O/stress_parity.log | cut -c1-200
python tools/stress_long.py 24 > $O/stress_long.log 2>&1; tail -2 $O/stress_long.log | cut -c1-200
