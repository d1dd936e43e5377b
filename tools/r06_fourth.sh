#!/bin/bash
# Round 6, fourth GPU call: the Cornell kernel making its own units' samples (generate_unit, SSX_FUSE_GEN=1): parity, A/B, traffic
O=gpurun_out/r06; mkdir -p $O
export SSX_DEBUG_ENV=1
SSX_FUSE_GEN=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py -m gpu -q -rf > $O/pytest_fuse_unit.log 2>&1; echo "pytest (SSX_FUSE_GEN=1) rc=$?"; grep -E "^FAILED|^ERROR| passed| failed" $O/pytest_fuse_unit.log | cut -c1-300
python -m pytest tests/test_bench_multi.py -m gpu -q -k dry_run 2>&1 | tail -2
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["stage_ms"], d["ranks"][0]["device_scratch_bytes"])'
for round in 1 2 3; do
	for SW in 0 1; do
		SSX_FUSE_GEN=$SW python bench.py --steps 10 --warmup 2 --quick 2>/dev/null | python -c "$P" "cornell fuse_unit=$SW"
	done
done
for SW in 0 1; do
	SSX_FUSE_GEN=$SW python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/cornell_fuse$SW.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/cornell_fuse$SW.json')); r=d['roofline']; print('fuse_unit=$SW', d['value'], r['traffic'], r['traffic_detail']['bytes_per_launch'], r['hbm']['traffic_bytes_per_sample'], d['check']['differing_floats'])"
done
SSX_FUSE_GEN=1 python bench.py --steps 6 --warmup 2 --quick --res 2048 --spp 64 --observer 2006 2>/dev/null | python -c "$P" "cornell 2048 cie2006 fuse_unit=1"
SSX_FUSE_GEN=0 python bench.py --steps 6 --warmup 2 --quick --res 2048 --spp 64 --observer 2006 2>/dev/null | python -c "$P" "cornell 2048 cie2006 fuse_unit=0"
