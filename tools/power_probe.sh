#!/bin/bash
# Samples rocm-smi power / clocks while bench.py runs (GPU box): is the path kernel running at its power cap?
R=$(pwd)
python $R/bench.py --steps 60 --warmup 5 --quick > $R/gpurun_out/power_bench.json 2>/dev/null &
B=$!
sleep 6
for i in $(seq 1 12); do
	/opt/rocm/bin/rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -i "power\|sclk\|GPU use" | tr '\n' ' '; echo
	sleep 0.5
done
wait $B
/opt/rocm/bin/rocm-smi --showmaxpower 2>/dev/null | grep -i "max\|power" | head -3
cut -c60-170 $R/gpurun_out/power_bench.json
