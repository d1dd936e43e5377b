#!/bin/bash
# Round 6, second GPU call: the formal suite 10 x with the per-session JIT cache (expect 10 green), the whole GPU suite, plane A/B of the black-surface shortcut.
O=gpurun_out/r06; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/pytest_gpu_1.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu_1.log | cut -c1-300
export SSX_DEBUG_ENV=1
for round in 1 2; do
	for SW in 1 0; do
		SSX_BLACK_SHORTCUT=$SW python bench.py --steps 6 --warmup 2 --quick --scene plane-srgb --res 1024 --spp 1024 --scratch-cap-gb 64 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('plane shortcut=$SW cap64', d['value'], d['ms_per_step'], d['roofline']['stage_ms'])"
	done
done
python bench.py --steps 6 --warmup 2 --quick --scene plane-srgb --res 1024 --spp 1024 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('plane default cap (8 GB)', d['value'], d['ms_per_step'], d['roofline']['stage_ms'], d['ranks'][0]['device_scratch_bytes'])"
python bench.py --steps 6 --warmup 2 --quick --scene plane-srgb --res 1024 --spp 1024 --scratch-cap-gb 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('plane cap 20 GB', d['value'], d['ms_per_step'], d['roofline']['stage_ms'], d['ranks'][0]['device_scratch_bytes'])"
python bench.py --steps 10 --warmup 2 --quick 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cornell', d['value'], d['ms_per_step'], d['roofline']['stage_ms'])"
python bench.py --dist-dry-run > $O/dry_run_n1.json 2> $O/dry_run_n1.err; cut -c1-1500 $O/dry_run_n1.json
SSX_BENCH_FORCE_DIST=1 python bench.py --dist-dry-run > $O/dry_run_n1_dist.json 2> $O/dry_run_n1_dist.err; cut -c1-1500 $O/dry_run_n1_dist.json
bash tools/formal_repeat.sh 10 > $O/formal_repeat_after.log 2>&1; grep -E "passed|failed|green" $O/formal_repeat_after.log | cut -c1-200
