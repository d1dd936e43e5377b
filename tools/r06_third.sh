#!/bin/bash
# Round 6, third GPU call: parity with the fused sample generation, its A/B on plane-srgb, the quad-record stride ablation (LDS bank conflicts),
# the headline render in four launches (device scratch below 2 GB)
O=gpurun_out/r06; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu_2.log 2>&1; echo "pytest rc=$?"; grep -E "^FAILED|^ERROR| passed| failed" $O/pytest_gpu_2.log | cut -c1-300
export SSX_DEBUG_ENV=1
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["stage_ms"], d["ranks"][0]["device_scratch_bytes"])'
for round in 1 2 3; do
	for SW in 1 0; do
		SSX_FUSE_GEN=$SW python bench.py --steps 6 --warmup 2 --quick --scene plane-srgb --res 1024 --spp 1024 --scratch-cap-gb 20 2>/dev/null | python -c "$P" "plane fuse=$SW"
	done
done
bash tools/build_variant.sh quadpad4 -DSSX_QUAD_PAD_WORDS=4 | tail -1
bash tools/ab_bench.sh simple_spectral_amd/libssx_hip_quadpad4.so 2>&1 | cut -c1-200
bash tools/pmc_lds.sh "" simple_spectral_amd/libssx_hip_quadpad4.so 2>&1 | grep -v generate
for round in 1 2 3; do
	python bench.py --steps 10 --warmup 2 --quick 2>/dev/null | python -c "$P" "cornell one launch"
	python bench.py --steps 10 --warmup 2 --quick --batch 64 2>/dev/null | python -c "$P" "cornell batch 64"
	python bench.py --steps 10 --warmup 2 --quick --batch 128 2>/dev/null | python -c "$P" "cornell batch 128"
done
