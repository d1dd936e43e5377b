#!/bin/bash
# Round 6, fourteenth GPU call: long fuzz on the final tree: warped built-in scenes (topology-specialised kernels) and random scenes (generic / compiled at upload)
O=gpurun_out/r06g; mkdir -p $O
timeout 1500 python tools/fuzz_scenes.py --warped 220000 ${1:-12000} > $O/fuzz_warped_long.log 2>&1; echo "warped rc=$?"; tail -2 $O/fuzz_warped_long.log | cut -c1-500
timeout 900 python tools/fuzz_scenes.py 170000 ${2:-6000} > $O/fuzz_scenes_more.log 2>&1; echo "random rc=$?"; tail -2 $O/fuzz_scenes_more.log | cut -c1-500
