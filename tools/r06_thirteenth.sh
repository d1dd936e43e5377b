#!/bin/bash
# Round 6, thirteenth GPU call: the parity suites with the generic kernel forced, again (the closing run's red there was the new warped-scene test
# asserting the topology kernel under a forced generic one: fixed in the test)
export SSX_DEBUG_ENV=1
SKIP="not specialised_for_builtin and not compiled_at_upload and not shadow_queue_layouts and not camera_rays_pretraced and not calibration_and_device_scratch and not background and not two_threads"
echo "== SSX_GENERIC_KERNEL=1 (rerun, tools/r06_thirteenth.sh)"; env SSX_GENERIC_KERNEL=1 timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py -q -m gpu -rf -k "$SKIP" 2>&1 | grep -E "^(FAILED|ERROR)|^E  | passed| failed| error" | cut -c1-300
