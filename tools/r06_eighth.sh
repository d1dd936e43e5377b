#!/bin/bash
# Round 6, eighth GPU call: the run-time unit grab against the previous revision's library (built in the build container: libssx_hip_prev.so), parity
O=gpurun_out/r06; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py tests/test_gpu_variants.py -m gpu -q -rf > $O/pytest_grab.log 2>&1; echo "pytest rc=$?"; grep -E "^FAILED|^ERROR| passed| failed" $O/pytest_grab.log | cut -c1-300
export SSX_DEBUG_ENV=1
bash tools/ab_bench.sh simple_spectral_amd/libssx_hip_prev.so 2>&1 | cut -c1-170
BENCH_ARGS="--scene plane-srgb --res 1024 --spp 1024 --scratch-cap-gb 20" bash tools/ab_bench.sh simple_spectral_amd/libssx_hip_prev.so 2>&1 | cut -c1-170
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["stage_ms"])'
for G in 1 2 4 8; do SSX_UNIT_GRAB=$G python bench.py --steps 6 --warmup 2 --quick --scene plane-srgb --res 1024 --spp 1024 --scratch-cap-gb 20 2>/dev/null | python -c "$P" "plane grab=$G"; done
for G in 1 2 4; do SSX_UNIT_GRAB=$G python bench.py --steps 10 --warmup 2 --quick 2>/dev/null | python -c "$P" "cornell grab=$G"; done
