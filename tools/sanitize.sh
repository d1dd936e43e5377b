#!/bin/bash
# Sanitizer pass over the host side (SURVEY.md section 5 "race detection / sanitizers"; the reference's hazard it mirrors:
# src/renderer.hpp:52, a `volatile bool` shared between threads).  Works in a scratch directory -- nothing sanitized ever lands in the
# package or travels to a GPU box as the product:
#   A. -fsanitize=address,undefined: libssx_host.so, the CLI, the oracle and tests/ddmath_host.cpp rebuilt with it in a scratch copy of
#      the tree; the CPU suite runs on them (LD_PRELOAD of the runtime into an unsanitized python; the torch.distributed test is left
#      out: an ASan-preloaded torch is not what is under test), and with a GPU the CLI / host-ABI GPU tests;
#   B. tests/sanitize_host.cpp -- the threaded host code in one executable: scene + colour tables from six threads, the Jakob-Hanika
#      fitter's pool, the oracle's tile-queue workers, and with a GPU the C++ host Renderer's start / progress / stop / wait -- built with
#      -fsanitize=address,undefined and with -fsanitize=thread (no interpreter in that process: python under a preloaded TSan deadlocks;
#      run with ASLR off, `setarch -R`, which this TSan needs on kernels with 32-bit mmap entropy).
# usage: tools/sanitize.sh [--gpu-only] [logfile]     exit code 0 = no report.  --gpu-only: the harness (A's suites run anywhere).
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
GPU_ONLY=0; if [ "${1:-}" = "--gpu-only" ]; then GPU_ONLY=1; shift; fi
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
ASAN=$(gcc -print-file-name=libasan.so); UBSAN=$(gcc -print-file-name=libubsan.so)
HOSTSRC="simple_spectral_amd/host/spectrum.cpp simple_spectral_amd/host/color.cpp simple_spectral_amd/host/jh2019.cpp simple_spectral_amd/host/meng2015.cpp simple_spectral_amd/host/scene.cpp simple_spectral_amd/host/image_io.cpp simple_spectral_amd/host/renderer.cpp simple_spectral_amd/host/host_api.cpp"
ORCSRC="oracle/oracle_color.c oracle/oracle_math.c oracle/oracle_scene.c oracle/oracle_render.c"
FAIL=0
harness() { # $1 = name, $2 = sanitizer flags: builds and runs tests/sanitize_host.cpp with the host sources and the oracle's
	echo "== harness tests/sanitize_host.cpp: $2"
	for f in $ORCSRC; do gcc -O1 -g -std=gnu11 -ffp-contract=off -fno-fast-math $2 -c $f -o /tmp/$$_$(basename $f).o || return 1; done
	g++ -O1 -g -std=c++17 -ffp-contract=off $2 tests/sanitize_host.cpp $HOSTSRC /tmp/$$_oracle_*.o -o $S/sanitize_host_$1 -lz -ldl -lpthread -lm || return 1
	rm -f /tmp/$$_oracle_*.o
	( cd "$S" && setarch "$(uname -m)" -R ./sanitize_host_$1 $([ $HAVE_GPU = 1 ] && echo gpu) ) 2>&1 | tail -12
	return ${PIPESTATUS[0]}
}
{
echo "== sanitize.sh $(date -u +%FT%TZ)  gcc $(gcc -dumpversion)  gpu=$HAVE_GPU"
cd "$S"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=66:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1:exitcode=67
# (the ROCm runtime is not instrumented: TSan cannot see its own synchronisation and reports its internal hand-overs as races; what is
# under test is the host code's threads, so reports whose stacks lie in those libraries are suppressed -- the file is written here)
printf 'race:libhsa-runtime64.so\nrace:libamdhip64.so\ncalled_from_lib:libhsa-runtime64.so\ncalled_from_lib:libamdhip64.so\ncalled_from_lib:libamd_comgr.so\nrace:libamd_comgr.so\n' > "$S/tsan.supp"
export TSAN_OPTIONS=halt_on_error=1:exitcode=68:report_signal_unsafe=0:ignore_noninstrumented_modules=1:suppressions=$S/tsan.supp
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined"
if [ $GPU_ONLY = 0 ]; then
	# ------------------------------------------------------------ A. the suites on sanitized libraries
	echo "== build: $SAN"
	SSX_HOST_EXTRA_FLAGS="$SAN -g" python -c "from simple_spectral_amd import build as b; b.build_host(force=True, verbose=False)" || FAIL=1
	make -C oracle -s clean >/dev/null; make -C oracle -s CFLAGS="-O1 -g -std=gnu11 -fPIC -ffp-contract=off -fno-fast-math $SAN" || FAIL=1
	[ -d /root/reference/src ] && make -C oracle -s ref
	sed -i 's/"g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"/"g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"/' tests/test_fmath.py
	echo "== CPU suite under ASan + UBSan (LD_PRELOAD: the sanitized libraries are loaded by an unsanitized python)"
	LD_PRELOAD="$ASAN $UBSAN" timeout 3000 python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider --deselect tests/test_distributed_cpu.py 2>&1 | tail -6
	[ ${PIPESTATUS[0]} -eq 0 ] || FAIL=1
	echo "== CLI binary (ASan + UBSan linked in): usage and argument errors"
	for A in "" "--scene=nope --width=8 --height=8 --samples=1 --output=/tmp/x.png" "--scene=cornell --width=0 --height=8 --samples=1 --output=/tmp/x.png" "--scene=cornell --width=8"; do
		./simple-spectral $A > /tmp/san_cli.out 2>&1; RC=$?; echo "  simple-spectral $A -> rc $RC: $(head -1 /tmp/san_cli.out | cut -c1-90)"; grep -E "ERROR: AddressSanitizer|runtime error" /tmp/san_cli.out && FAIL=1
	done
	if [ $HAVE_GPU = 1 ]; then
		echo "== GPU box: CLI + host-ABI tests with the sanitized host library next to the real libssx_hip.so"
		LD_PRELOAD="$ASAN $UBSAN" timeout 1500 python -m pytest tests/test_cli.py tests/test_host_and_abi.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -4
		[ ${PIPESTATUS[0]} -eq 0 ] || FAIL=1
	fi
fi
# ------------------------------------------------------------ B. the threaded host code in one sanitized executable
harness asan "$SAN" || FAIL=1
harness tsan "-fsanitize=thread -fno-omit-frame-pointer" || FAIL=1
echo "== result: $([ $FAIL = 0 ] && echo 'no sanitizer report' || echo 'REPORTS / FAILURES above')"
} 2>&1 | tee "$LOG"
grep -q "result: no sanitizer report" "$LOG"
