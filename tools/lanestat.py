#!/usr/bin/env python3
"""Lane occupancy per region of the path megakernel (GPU box).  Builds a profiling copy of the library
with -DSSX_LANESTAT (simple_spectral_amd/csrc/ssx_lanestat.h) next to the product library, renders the
bench workload with it and prints, per region, entries, mean active lanes and occupancy.
    python tools/lanestat.py [--scene cornell-srgb --res 512 --spp 64]"""
import argparse, ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = {0: "primary trace: lanes with a ray", 1: "primary pass-2 trips", 2: "shadow trace: lanes with a ray", 3: "shadow pass-2 trips", 4: "path_step: lanes shading a hit",
         5: "emission lookup", 6: "albedo: texture", 7: "albedo: constant", 8: "light sampling", 9: "NEE contribution", 10: "shadow ray parked",
         11: "BSDF sample", 12: "continue (store fs/np)", 13: "iteration: lanes with a path", 14: "fold level x way", 15: "flux -> XYZ", 16: "camera trace: lanes with a ray", 17: "camera pass-2 trips", 18: "black surface: draws only", 19: "refill: lanes taking a sample"}
ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="cornell-srgb"); ap.add_argument("--res", type=int, default=512); ap.add_argument("--spp", type=int, default=64)
args = ap.parse_args()
lib = os.path.join(ROOT, "gpurun_out", "libssx_hip_lanestat.so")
os.makedirs(os.path.dirname(lib), exist_ok=True)
from simple_spectral_amd import build as b
subprocess.check_call([b.hipcc()] + b.HIP_FLAGS + ["-DSSX_LANESTAT"] + b.HIP_SRC + ["-o", lib, "-lpthread"])
os.environ["SSX_DEBUG_ENV"] = "1"
os.environ["SSX_HIP_LIB_OVERRIDE"] = lib
import torch
from simple_spectral_amd import Options, Renderer, _capi
r = Renderer(Options(scene_name=args.scene, res=(args.res, args.res), spp=args.spp, texture="crystal-lizard-512.png"))
h = _capi.hip_lib()
out = (C.c_ulonglong * 40)()
h.ssx_lanestat(out, 1)
buf = torch.zeros((args.res, args.res, 4), device="cuda")
r.render_device(buf.data_ptr(), torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
h.ssx_lanestat(out, 0)
samples = args.res * args.res * args.spp
print("%-36s %14s %12s %10s %14s" % ("region", "entries", "lanes/entry", "occupancy", "lanes/sample"))
for k in range(20):
    lanes, cnt = out[2 * k], out[2 * k + 1]
    if cnt:
        print("%-36s %14d %12.2f %9.1f%% %14.3f" % (NAMES.get(k, str(k)), cnt, lanes / cnt, 100.0 * lanes / cnt / 64.0, lanes / samples))
