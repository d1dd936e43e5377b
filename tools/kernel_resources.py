#!/usr/bin/env python3
"""Registers, spills, scratch and static LDS of every kernel in csrc/ssx_api.hip (hipcc -S, no GPU needed).
    python tools/kernel_resources.py [--probe] [-D MACRO ...]      --probe: csrc/ssx_kernels.hip alone with -DSSX_PROBE_BUILD (generic and
    Cornell path kernels only: a third of the compile time, for register-pressure experiments)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simple_spectral_amd import build as _b
_b.embed_sources()
probe = "--probe" in sys.argv
if probe:
    sys.argv.remove("--probe")
with tempfile.TemporaryDirectory() as td:
    asm = os.path.join(td, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-S", "--cuda-device-only", "-o", asm,
                           os.path.join(ROOT, "simple_spectral_amd", "csrc", "ssx_kernels.hip" if probe else "ssx_api.hip")] + (["-DSSX_PROBE_BUILD"] if probe else []) + sys.argv[1:], stderr=subprocess.DEVNULL)
    t = open(asm).read()
print("%-28s %6s %6s %6s %6s %8s" % ("kernel", "vgpr", "vspill", "sgpr", "sspill", "scratch"))
for b in t.split("  - .agpr_count:")[1:]:
    g = lambda k: re.search(r"\." + k + r":\s+(\S+)", b).group(1)
    print("%-28s %6s %6s %6s %6s %8s" % (g("name"), g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"), g("private_segment_fixed_size")))
