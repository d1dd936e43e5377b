#!/bin/bash
# Round 6, fifth GPU call: plane-srgb pass 1 culls the triangles behind the ray's origin -- parity on the plane cases, A/B against a build without it
O=gpurun_out/r06; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py -m gpu -q -rf -k "plane or black or rgb or per_sample or config4 or variants or mirror or without" > $O/pytest_cull.log 2>&1; echo "pytest rc=$?"; grep -E "^FAILED|^ERROR| passed| failed" $O/pytest_cull.log | cut -c1-300
export SSX_DEBUG_ENV=1
bash tools/build_variant.sh nocull -DSSX_NO_BEHIND_CULL | tail -1
BENCH_ARGS="--scene plane-srgb --res 1024 --spp 1024 --scratch-cap-gb 20" bash tools/ab_bench.sh simple_spectral_amd/libssx_hip_nocull.so 2>&1 | cut -c1-160
python tools/lanestat.py --scene plane-srgb --res 1024 --spp 64 > $O/plane_lanestat_after.log 2>&1; tail -22 $O/plane_lanestat_after.log
