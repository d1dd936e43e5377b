P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["stage_ms"], d["roofline"]["device_scratch_bytes"])'
for r in 1 2; do for b in 0 128 64 32; do python bench.py --steps 10 --warmup 2 --no-cpu-baseline --batch $b 2>/dev/null | python -c "$P" batch$b; done; done
