#!/bin/bash
# Sanitizer pass over the host side (SURVEY.md section 5 "race detection / sanitizers"; the reference's hazard it mirrors:
# src/renderer.hpp:52, a `volatile bool` shared between threads).  Works on a scratch copy of the tree -- nothing sanitized ever lands
# in the package or travels to a GPU box as the product:
#   1. libssx_host.so, the CLI, the oracle and tests/ddmath_host.cpp built with -fsanitize=address,undefined; the CPU suite (minus the
#      torch.distributed test: an ASan-preloaded torch is not what is under test) and, when a GPU is present, the CLI / host-ABI GPU tests;
#   2. the same host code built with -fsanitize=thread: the threaded table preparation (Color::init), the Jakob-Hanika fitter's thread pool,
#      the oracle's tile-queue renderer (a port of the reference's worker loop) and -- with a GPU -- the C++ host Renderer's start/stop/wait.
# usage: tools/sanitize.sh [logfile]        (exit code 0 = no report)
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
LOG=${1:-$R/gpurun_out/sanitize.log}
mkdir -p "$(dirname "$LOG")"
S=$(mktemp -d /tmp/ssx_san.XXXXXX)
trap 'rm -rf "$S"' EXIT
tar -C "$R" --exclude=.git --exclude=gpurun_out --exclude='*.so' --exclude=simple-spectral --exclude=__pycache__ --exclude=profiles -cf - . | tar -C "$S" -xf -
cp "$R"/simple_spectral_amd/libssx_hip.so "$S"/simple_spectral_amd/ 2>/dev/null   # the HIP library itself is not host code: taken as built
HAVE_GPU=0; python - <<'PY' 2>/dev/null && HAVE_GPU=1
import torch, sys
sys.exit(0 if torch.cuda.is_available() else 1)
PY
ASAN=$(gcc -print-file-name=libasan.so); UBSAN=$(gcc -print-file-name=libubsan.so); TSAN=$(gcc -print-file-name=libtsan.so)
FAIL=0
{
echo "== sanitize.sh $(date -u +%FT%TZ)  gcc $(gcc -dumpversion)  gpu=$HAVE_GPU"
cd "$S"
# ---------------------------------------------------------------- 1. address + undefined behaviour
echo "== build: -fsanitize=address,undefined"
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -g"
SSX_HOST_EXTRA_FLAGS="$SAN" python -c "from simple_spectral_amd import build as b; b.build_host(force=True, verbose=False)" || FAIL=1
make -C oracle -s clean >/dev/null; make -C oracle -s CFLAGS="-O1 -std=gnu11 -fPIC -ffp-contract=off -fno-fast-math $SAN" || FAIL=1
[ -d /root/reference/src ] && make -C oracle -s ref
sed -i 's/"g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"/"g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"/' tests/test_fmath.py
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=66:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1:exitcode=67
echo "== CPU suite under ASan + UBSan (LD_PRELOAD: the sanitized libraries are loaded by an unsanitized python)"
LD_PRELOAD="$ASAN $UBSAN" timeout 3000 python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider --deselect tests/test_distributed_cpu.py 2>&1 | tail -15
[ ${PIPESTATUS[0]} -eq 0 ] || FAIL=1
echo "== CLI binary (ASan + UBSan linked in): argument errors, unknown scene, help"
for A in "--help" "--scene nope -w 8 -h 8 -spp 1 --output /tmp/x.png" "-w 0" "--spp"; do ./simple-spectral $A > /tmp/san_cli.out 2>&1; RC=$?; echo "  simple-spectral $A -> rc $RC"; grep -E "ERROR: AddressSanitizer|runtime error" /tmp/san_cli.out && FAIL=1; done
if [ $HAVE_GPU = 1 ]; then
	echo "== GPU box: CLI + host-ABI tests with the sanitized host library next to the real libssx_hip.so"
	LD_PRELOAD="$ASAN $UBSAN" timeout 1500 python -m pytest tests/test_cli.py tests/test_host_and_abi.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -8
	[ ${PIPESTATUS[0]} -eq 0 ] || FAIL=1
fi
# ---------------------------------------------------------------- 2. thread sanitizer
echo "== build: -fsanitize=thread"
SAN="-fsanitize=thread -fno-omit-frame-pointer -g"
SSX_HOST_EXTRA_FLAGS="$SAN" python -c "from simple_spectral_amd import build as b; b.build_host(force=True, verbose=False)" || FAIL=1
make -C oracle -s clean >/dev/null; make -C oracle -s CFLAGS="-O1 -std=gnu11 -fPIC -ffp-contract=off -fno-fast-math $SAN" || FAIL=1
export TSAN_OPTIONS=halt_on_error=1:exitcode=68:report_signal_unsafe=0
echo "== threaded host code under TSan: Color::init + scene tables (ssh_scene_create), the Jakob-Hanika fitter's pool, the oracle's tile-queue workers"
LD_PRELOAD="$TSAN" timeout 3000 python -m pytest tests/test_host_and_abi.py tests/test_oracle_pins.py tests/test_unit_cases_cpu.py -x -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -8
[ ${PIPESTATUS[0]} -eq 0 ] || FAIL=1
LD_PRELOAD="$TSAN" timeout 600 python - <<'PY' 2>&1 | tail -5
import sys; sys.path.insert(0, "tests")
import numpy as np, oracle_lib as ol
o = ol.Oracle("cornell-srgb", texture="test-img.png")
a = o.render(48, 40, 3, nthreads=8); b = o.render(48, 40, 3, nthreads=1)
assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
print("oracle tile queue, 8 workers against 1: same image, no TSan report")
PY
[ ${PIPESTATUS[0]} -eq 0 ] || FAIL=1
if [ $HAVE_GPU = 1 ]; then
	echo "== GPU box: the C++ host Renderer (worker thread, stop flag, progress) through the CLI tests under TSan"
	LD_PRELOAD="$TSAN" timeout 1500 python -m pytest tests/test_cli.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -8
	[ ${PIPESTATUS[0]} -eq 0 ] || FAIL=1
fi
echo "== result: $([ $FAIL = 0 ] && echo 'no sanitizer report' || echo 'REPORTS / FAILURES above')"
} 2>&1 | tee "$LOG"
grep -q "result: no sanitizer report" "$LOG"
