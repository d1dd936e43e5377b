#!/usr/bin/env python3
"""Where the path kernel's waves spend their TIME, measured (GPU box).  Builds the profiling copy of the library (-DSSX_REGTIME:
csrc/ssx_lanestat.h SSX_TIME -- the shader clock between wave-uniform marks of the path loop, summed per region over all waves),
renders the bench workload with it and prints each region's share of the waves' time next to its share of the VALU issue cycles
as tools/isa_census.py models them (--census FILE: its csv).  A wave's time in a region = its own issue cycles + the cycles it
waited (for the SIMD's other three waves, for LDS / HBM round trips): a region whose time share exceeds its issue share is where
latency, not issue, sets the pace.
    python tools/regtime.py [--scene cornell-srgb --res 512 --spp 64] [--census profiles/r05/isa_census.csv]"""
import argparse, csv, ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = {3: "shade: path_step + end_path (loop back included)", 4: "rotate / unit fetch", 12: "shadow trace: queue read, ray set-up, pass 1", 13: "shadow trace: pass 2",
         5: "shadow flush: result store", 6: "fold (unit_fold)", 14: "flush_fold: tests", 7: "primary trace: ray set-up, pass 1", 8: "primary trace: pass 2",
         9: "after trace: hit -> path state, st", 10: "refill"}
# regions of tools/isa_census.py that make up each timed region
CENSUS = {3: ("light:", "albedo:", "bsdf:", "nee:", "path_step:", "log level", "loop: end_path", "emission", "loop: other", "cold"), 4: ("loop: rotate", "loop: unit fetch"),
          12: ("trace shadow: ray_setup", "trace shadow: pass 1", "trace shadow: masks"), 13: ("trace shadow: pass 2",), 5: ("shadow flush",), 6: ("fold:",), 14: ("loop: flush_fold",),
          7: ("trace primary: ray_setup", "trace primary: pass 1", "trace primary: masks"), 8: ("trace primary: pass 2",), 9: ("loop: after trace", "loop: hit st"), 10: ("loop: refill",)}
ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="cornell-srgb"); ap.add_argument("--res", type=int, default=512); ap.add_argument("--spp", type=int, default=64)
ap.add_argument("--census", default="")
args = ap.parse_args()
lib = os.path.join(ROOT, "gpurun_out", "libssx_hip_regtime.so")
os.makedirs(os.path.dirname(lib), exist_ok=True)
from simple_spectral_amd import build as b
b.embed_sources()
subprocess.check_call([b.hipcc()] + b.HIP_FLAGS + ["-DSSX_REGTIME"] + b.HIP_SRC + ["-o", lib, "-lpthread", "-ldl"])
os.environ["SSX_DEBUG_ENV"] = "1"
os.environ["SSX_HIP_LIB_OVERRIDE"] = lib
import torch
from simple_spectral_amd import Options, Renderer, _capi
r = Renderer(Options(scene_name=args.scene, res=(args.res, args.res), spp=args.spp, texture="crystal-lizard-512.png"))
h = _capi.hip_lib()
out = (C.c_ulonglong * 16)()
buf = torch.zeros((args.res, args.res, 4), device="cuda")
r.render_device(buf.data_ptr(), torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()   # warm-up
h.ssx_regtime(out, 1)
r.set_timing(True)
r.render_device(buf.data_ptr(), torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
ms = r.get_timing()["path"]
h.ssx_regtime(out, 0)
total = float(sum(out))
issue = {}
if args.census and os.path.exists(args.census):
    rows = list(csv.DictReader(open(args.census)))
    tot_c = sum(float(x["cycles_per_iteration"]) for x in rows)
    for k, pre in CENSUS.items():
        issue[k] = sum(float(x["cycles_per_iteration"]) for x in rows if x["region"].startswith(pre)) / tot_c
print("path kernel of the profiling build: %.2f ms for %d samples; %.3g shader-clock cycles summed over the waves' regions" % (ms, args.res * args.res * args.spp, total))
print("%-52s %9s %9s %s" % ("region", "time", "issue", "time / issue"))
for k in (3, 7, 8, 12, 13, 5, 6, 14, 9, 10, 4):
    t = out[k] / total
    i = issue.get(k)
    print("%-52s %8.1f%% %9s %s" % (NAMES[k], 100 * t, ("%8.1f%%" % (100 * i)) if i is not None else "", ("%.2f" % (t / i)) if i else ""))
