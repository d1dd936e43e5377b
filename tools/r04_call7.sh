#!/bin/bash
O=gpurun_out/r04_g; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_cli.py tests/test_gpu_units.py -m gpu -q -k "tile_major or proved_against or abort" > $O/pytest_new.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_new.log | cut -c1-300
bash tools/profile_round.sh r04mid > $O/profile_round.log 2>&1; tail -4 $O/profile_round.log | cut -c1-300
