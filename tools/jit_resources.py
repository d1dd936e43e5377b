#!/usr/bin/env python3
"""Registers / spills of the path kernels compiled around a pass 1 generated for a scene's own mesh topology
(csrc/ssx_jit.h: what hipRTC builds at upload), checked OFFLINE with hipcc -S -- no GPU needed.  The scene is the Cornell box
with one corner of quad 0 moved apart from its twins (tests/test_gpu_parity.py::test_pass1_compiled_at_upload_for_any_topology).
    python tools/jit_resources.py [-DMACRO ...]"""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from simple_spectral_amd import _capi, build as b
from simple_spectral_amd.renderer import Scene

b.embed_sources()
SC = Scene("cornell-srgb", texture=os.path.join(ROOT, "data", "scenes", "test-img.png"))
d = SC.desc.contents
ids, vid = {}, []
for q in range(d.n_quads):
    Q = d.quads[q]
    for k, v in enumerate((Q.v00, Q.v10, Q.v11, Q.v01)):
        key = tuple(v.pos) if not (q == 0 and k == 0) else ("moved",)
        vid.append(ids.setdefault(key, len(ids)))
vid = np.array(vid, dtype=np.uint8)
lib = C.CDLL(b.HIP_LIB)
lib.ssx_debug_pass1_source.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_char_p, C.c_size_t]
buf = C.create_string_buffer(1 << 20)
n = lib.ssx_debug_pass1_source(vid.ctypes.data, d.n_quads, b"jit", buf, len(buf))
assert n > 0, n
with tempfile.TemporaryDirectory() as td:
    open(os.path.join(td, "ssx_pass1_jit.h"), "wb").write(buf.value)
    for name, path in b.EMBED:
        if not path.endswith(".hip"):
            open(os.path.join(td, os.path.basename(path)), "w").write(open(path).read())
    src = os.path.join(td, "ssx_kernels.hip")
    open(src, "w").write(open(os.path.join(ROOT, "simple_spectral_amd", "csrc", "ssx_kernels.hip")).read())
    asm = os.path.join(td, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-DSSX_JIT_BUILD",
                           "-S", "--cuda-device-only", "-o", asm, src] + sys.argv[1:])
    t = open(asm).read()
print("%-28s %6s %6s %6s %6s %8s" % ("kernel", "vgpr", "vspill", "sgpr", "sspill", "scratch"))
for blk in t.split("  - .agpr_count:")[1:]:
    g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)
    print("%-28s %6s %6s %6s %6s %8s" % (g("name"), g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"), g("private_segment_fixed_size")))
