"""Build the two in-tree shared libraries (no JIT cache: the .so files travel with the repo).

    libssx_hip.so   HIP kernels + C ABI (include/ssx.h), hipcc --offload-arch=gfx950
    libssx_host.so  host-side table / scene / image code (include/ssx_host.h), g++
"""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
HIP_LIB = os.path.join(PKG, "libssx_hip.so")
# The same library with the inter-wave hand-over of the pixel sums stated in the HIP memory model (csrc/ssx_kernels.hip unit_fold:
# acquire / release agent-scope read-modify-writes instead of relaxed atomics + s_waitcnt; 25 % slower).  Not what runs by default:
# tests/test_gpu_variants.py runs it next to the default build on every `pytest -m gpu` (stress cases, fuzzed scenes and the whole
# parity suites), which is what validates the default build's below-the-model ordering on the box and toolchain at hand.  Built on
# request only (build_hip(formal=True): __graft_entry__.build(), `python -m simple_spectral_amd.build --variants`, SSX_BUILD_FORMAL=1).
HIP_LIB_FORMAL = os.path.join(PKG, "libssx_hip_formal.so")
HOST_LIB = os.path.join(PKG, "libssx_host.so")

HIP_SRC = [os.path.join(PKG, "csrc", f) for f in ("ssx_api.hip",)]
HIP_DEPS = [os.path.join(PKG, "csrc", f) for f in ("ssx_api.hip", "ssx_kernels.hip", "ssx_debug.hip", "ssx_blob.h", "ssx_exact.h", "ssx_lanestat.h", "ssx_pass1_gen.h", "ssx_jit.h", "ssx_ddmath.h")] + [
    os.path.join(ROOT, "include", f) for f in ("ssx.h", "ssx_fmath.h")]
HOST_SRC = [os.path.join(PKG, "host", f) for f in
            ("spectrum.cpp", "color.cpp", "jh2019.cpp", "meng2015.cpp", "scene.cpp", "image_io.cpp", "renderer.cpp", "host_api.cpp")]
HOST_DEPS = HOST_SRC + [os.path.join(PKG, "host", f) for f in
                        ("spectrum.hpp", "color.hpp", "jh2019.hpp", "scene.hpp", "image_io.hpp", "renderer.hpp")] + [
    os.path.join(ROOT, "include", f) for f in ("ssx.h", "ssx_host.h")]
CLI_SRC = os.path.join(PKG, "host", "main.cpp")
CLI_BIN = os.path.join(ROOT, "simple-spectral")

# -ffp-contract=off is part of the numerics contract (bit parity with the oracle).
# -fno-slp-vectorize: the SLP vectoriser pairs adjacent scalar f32 adds / multiplies into v_pk_add_f32 / v_pk_mul_f32.  On
# gfx950 a packed op issues in 4.4 cycles against 2 x 2.5 for the VOP2 forms and stalls the issue of what follows (measured:
# SQ_WAIT_INST_ANY -28 % without them); A/B on one box: +4.7 % (2763 against 2640 Msamples/s).  Same operations, same bits.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared"]
HOST_FLAGS = ["-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall", "-Wextra"] + os.environ.get("SSX_HOST_EXTRA_FLAGS", "").split()  # (tools/sanitize.sh: -fsanitize=...)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


# The path kernel reads data that other lanes of the SAME wave stored, relying on gfx9 behaviour that is below
# what the HIP memory model promises (see "Memory-ordering contract" in csrc/ssx_kernels.hip): in-order
# vector-memory issue per wave, write-through L1, agent-scope loads served by the L2.  That was validated
# (bit-exact GPU parity tests, incl. the many-units-per-wave stress case) on gfx950 with the ROCm major
# versions listed here; another compiler generation has to pass `pytest -m gpu` before it is added.
TESTED_ROCM_MAJOR = ("7",)


def check_toolchain():
    root = os.path.dirname(os.path.dirname(os.path.realpath(hipcc())))
    version = ""
    for cand in (os.path.join(root, ".info", "version"), "/opt/rocm/.info/version"):
        if os.path.exists(cand):
            version = open(cand).read().strip()
            break
    # fail closed: a toolchain whose version cannot be read is as untested as one of another major version
    if (not version or version.split(".")[0] not in TESTED_ROCM_MAJOR) and os.environ.get("SSX_ALLOW_UNTESTED_ROCM") != "1":
        raise RuntimeError("ROCm %s is not a tested toolchain for libssx_hip.so (tested majors: %s): the kernel's same-wave memory "
                           "ordering relies on validated compiler/hardware behaviour.  Run the GPU parity tests and add the version to "
                           "simple_spectral_amd/build.py, or set SSX_ALLOW_UNTESTED_ROCM=1." % (version or "(version unreadable)", ", ".join(TESTED_ROCM_MAJOR)))
    return version


# The kernel sources as string literals, for the run-time specialisation of pass 1 (csrc/ssx_jit.h: hipRTC compiles the
# path kernels around a pass 1 generated for the uploaded scene's mesh topology).  Generated, not committed.
EMBED = (("kernels_hip", os.path.join(PKG, "csrc", "ssx_kernels.hip")), ("blob_h", os.path.join(PKG, "csrc", "ssx_blob.h")),
         ("exact_h", os.path.join(PKG, "csrc", "ssx_exact.h")), ("lanestat_h", os.path.join(PKG, "csrc", "ssx_lanestat.h")),
         ("fmath_h", os.path.join(ROOT, "include", "ssx_fmath.h")), ("pass1_gen_h", os.path.join(PKG, "csrc", "ssx_pass1_gen.h")))
SOURCES_GEN = os.path.join(PKG, "csrc", "ssx_sources_gen.h")


def embed_sources():
    out = ["// GENERATED by simple_spectral_amd/build.py from the files named below; do not edit, do not commit.", "#pragma once"]
    for name, path in EMBED:
        text = open(path).read()
        assert ')SSXSRC"' not in text
        out.append("// %s" % os.path.relpath(path, ROOT))
        # (string literals are limited to 64 KiB by the standard's minimum, not by clang; pieces keep other compilers happy)
        pieces = [text[i:i + 12000] for i in range(0, len(text), 12000)]
        out.append("static const char ssx_src_%s[] =\n%s;" % (name, "\n".join('R"SSXSRC(%s)SSXSRC"' % p for p in pieces)))
    text = "\n".join(out) + "\n"
    if not os.path.exists(SOURCES_GEN) or open(SOURCES_GEN).read() != text:
        open(SOURCES_GEN, "w").write(text)


def build_hip(force=False, verbose=False, formal=False):
    """libssx_hip.so; with formal=True (python -m simple_spectral_amd.build --variants, __graft_entry__.build(): what the GPU tests need)
    also libssx_hip_formal.so -- a second full compilation that only tests/test_gpu_variants.py loads, so not part of a plain build."""
    if force or _stale(HIP_LIB, HIP_DEPS):
        check_toolchain()
        embed_sources()
        cmd = [hipcc()] + HIP_FLAGS + HIP_SRC + ["-o", HIP_LIB, "-lpthread", "-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    if formal and (force or _stale(HIP_LIB_FORMAL, HIP_DEPS)):
        check_toolchain()
        embed_sources()
        cmd = [hipcc()] + HIP_FLAGS + ["-DSSX_ACCUM_FORMAL"] + HIP_SRC + ["-o", HIP_LIB_FORMAL, "-lpthread", "-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HIP_LIB


def build_host(force=False, verbose=False):
    srcs = [s for s in HOST_SRC if os.path.exists(s)]
    if force or _stale(HOST_LIB, HOST_DEPS):
        cmd = ["g++"] + HOST_FLAGS + ["-shared"] + srcs + ["-o", HOST_LIB, "-lz", "-ldl", "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    if os.path.exists(CLI_SRC) and (force or _stale(CLI_BIN, HOST_DEPS + [CLI_SRC])):
        cmd = ["g++"] + HOST_FLAGS + [CLI_SRC, "-o", CLI_BIN, "-L" + PKG, "-lssx_host",
                                      "-Wl,-rpath,$ORIGIN/simple_spectral_amd", "-ldl", "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HOST_LIB


def build_all(force=False, verbose=False, formal=False):
    return build_hip(force, verbose, formal), build_host(force, verbose)


if __name__ == "__main__":
    import sys
    if "--embed-only" in sys.argv:
        embed_sources()
    else:
        build_all(force="--force" in sys.argv, verbose=True, formal="--variants" in sys.argv or os.environ.get("SSX_BUILD_FORMAL") == "1")
