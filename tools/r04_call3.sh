#!/bin/bash
O=gpurun_out/r04_c; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log | cut -c1-300
python tools/jit_rate.py > $O/jit_rate.log 2>&1; cat $O/jit_rate.log | cut -c1-400
python tools/rank_share.py --configs headline --tag r04-stages > $O/rank_share_stages.log 2>&1
grep -o '"N": [0-9], "rank": [0-9].*' $O/rank_share_stages.log | cut -c1-260
python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_c/bench.json"))
print({k:d[k] for k in ("value","ms_per_step","value_device_resident","ms_per_step_device_resident")}, d["roofline"]["frac"], d["roofline"]["stage_ms"], d["roofline"]["hbm"]["traffic_over_algorithmic_8d"], d["cpu_baseline"]["value"], d["cpu_baseline"]["vs_reference_probe"])
PY
