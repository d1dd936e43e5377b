#!/bin/bash
# Round 6, twelfth GPU call: the warped built-in scenes (tests/crafted.py warped_builtin) on the topology-specialised kernels: the new GPU test, then the long fuzz
O=gpurun_out/r06f; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_units.py -q -m gpu -rf -k "warped or black_surfaces" > $O/pytest_warped.log 2>&1; echo "pytest rc=$?"; grep -E "^FAILED|^ERROR|^E  | passed| failed" $O/pytest_warped.log | tail -8 | cut -c1-300
timeout 1200 python tools/fuzz_scenes.py --warped 200000 ${1:-3000} > $O/fuzz_warped.log 2>&1; echo "fuzz rc=$?"; tail -3 $O/fuzz_warped.log | cut -c1-500
