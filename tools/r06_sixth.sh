#!/bin/bash
# Round 6, sixth GPU call: parity after the camera subexpressions moved to the host; the all-lanes release of one LDS word against one lane (LDS conflict attribution)
O=gpurun_out/r06; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py -m gpu -q -rf > $O/pytest_cam.log 2>&1; echo "pytest rc=$?"; grep -E "^FAILED|^ERROR| passed| failed" $O/pytest_cam.log | cut -c1-300
export SSX_DEBUG_ENV=1
bash tools/build_variant.sh rel1 -DSSX_RELEASE_ONE_LANE | tail -1
bash tools/ab_bench.sh simple_spectral_amd/libssx_hip_rel1.so 2>&1 | cut -c1-160
bash tools/pmc_lds.sh "" simple_spectral_amd/libssx_hip_rel1.so 2>&1 | grep -v generate
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["stage_ms"])'
for round in 1 2; do
python bench.py --steps 6 --warmup 2 --quick --scene plane-srgb --res 1024 --spp 1024 --scratch-cap-gb 20 2>/dev/null | python -c "$P" "plane"
python bench.py --steps 10 --warmup 2 --quick 2>/dev/null | python -c "$P" "cornell"
done
bash tools/pmc_lds.sh "--scene plane-srgb --res 1024 --spp 256" 2>&1 | grep -v generate
