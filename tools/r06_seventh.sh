#!/bin/bash
# Round 6, seventh GPU call: work units taken from the counter four at a time (-DSSX_UNIT_GRAB=4): A/B on both workloads, parity of the variant; scratch caps on plane-srgb
O=gpurun_out/r06; mkdir -p $O
export SSX_DEBUG_ENV=1
bash tools/build_variant.sh grab4 -DSSX_UNIT_GRAB=4 | tail -1
bash tools/build_variant.sh grab2 -DSSX_UNIT_GRAB=2 | tail -1
BENCH_ARGS="--scene plane-srgb --res 1024 --spp 1024 --scratch-cap-gb 20" bash tools/ab_bench.sh simple_spectral_amd/libssx_hip_grab4.so simple_spectral_amd/libssx_hip_grab2.so 2>&1 | cut -c1-170
bash tools/ab_bench.sh simple_spectral_amd/libssx_hip_grab4.so simple_spectral_amd/libssx_hip_grab2.so 2>&1 | cut -c1-170
SSX_HIP_LIB_OVERRIDE=$PWD/simple_spectral_amd/libssx_hip_grab4.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py -m gpu -q -rf > $O/pytest_grab4.log 2>&1; echo "pytest grab4 rc=$?"; grep -E "^FAILED|^ERROR| passed| failed" $O/pytest_grab4.log | cut -c1-300
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["stage_ms"], d["ranks"][0]["device_scratch_bytes"])'
for CAP in 8 16 20; do
python bench.py --steps 6 --warmup 2 --quick --scene plane-srgb --res 1024 --spp 1024 --scratch-cap-gb $CAP 2>/dev/null | python -c "$P" "plane cap $CAP"
done
