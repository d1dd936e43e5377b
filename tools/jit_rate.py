#!/usr/bin/env python3
"""Pass 1 specialised to a scene's own mesh topology, as a user meets it (GPU box).

  1. the built-in Cornell topology                                   (the yardstick)
  2. the same box with one corner moved apart from its twin, specialisation off: the generic loop
  3. that scene with the library's defaults, in a process with an empty cache: it starts on the generic kernel, the
     background thread compiles once 32 M samples have been rendered, the context switches kernels -- the rate per step shows when
  4. a second process: the code comes from the disk cache at upload

    python tools/jit_rate.py [--spp 256]        (SSX_CACHE_DIR is pointed at a scratch directory)"""
import argparse, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
ap = argparse.ArgumentParser(); ap.add_argument("--spp", type=int, default=256); ap.add_argument("--res", type=int, default=512)
ap.add_argument("--child", action="store_true")
a = ap.parse_args()
if not a.child:
    os.environ["SSX_CACHE_DIR"] = tempfile.mkdtemp(prefix="ssx-cache-")
import ctypes as C
import torch
import custom_scene as cs
from simple_spectral_amd import Options, Renderer, _capi
W = H = a.res
out = torch.zeros((H, W, 4), device="cuda"); s = torch.cuda.current_stream()


def step_ms(r, n=1):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): r.render_device(out.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def rate(r, label):
    step_ms(r, 3)
    ms = step_ms(r, 10)
    print("%-60s %-36s %8.1f Msamples/s  %.3f ms" % (label, r.plan_info()["pass1"], W * H * a.spp / ms / 1e3, ms), flush=True)
    return W * H * a.spp / ms


def moved():
    c = cs.CustomScene("cornell-srgb", texture="crystal-lizard-512.png")
    pos, st, m = c.quads[0]; pos = pos.copy(); pos[0, 0] += 1.0; c.quads[0] = (pos, st, m)
    return c, c.oracle()


o = dict(scene_name="cornell-srgb", res=(W, H), spp=a.spp, texture="crystal-lizard-512.png")
if a.child:
    t0 = time.time(); r = Renderer(Options(**o)); t_builtin = time.time() - t0
    c, orc = moved()
    t = time.time(); r.upload_scene_desc(c.desc(orc)); up = time.time() - t
    st = r.jit_status()[0]
    ca, cb = C.c_uint64(), C.c_uint64(); _capi.hip_lib().ssx_jit_counters(C.byref(ca), C.byref(cb))
    ms = step_ms(r, 3); ms = step_ms(r, 10)
    print(json.dumps({"upload_s": round(up, 4), "builtin_scene_create_s": round(t_builtin, 4), "state_after_upload": st, "compiled": ca.value, "disk_hits": cb.value,
                      "pass1": r.plan_info()["pass1"], "ms": round(ms, 3)}))
    sys.exit(0)

base = rate(Renderer(Options(**o)), "1. cornell-srgb (built-in topology)")
c, orc = moved()
r = Renderer(Options(jit_pass1=False, **o)); t = time.time(); r.upload_scene_desc(c.desc(orc)); up_generic = time.time() - t
gen = rate(r, "2. one corner moved, specialisation off (upload %.3f s)" % up_generic)
r = Renderer(Options(**o)); t = time.time(); r.upload_scene_desc(c.desc(orc)); up = time.time() - t
print("3. one corner moved, defaults, empty cache: upload %.3f s, state %d; ms per step:" % (up, r.jit_status()[0]), flush=True)
t0 = time.time(); trace = []; switched = None
while time.time() - t0 < 60:
    ms = step_ms(r)
    trace.append((round(time.time() - t0, 2), round(ms, 2), r.plan_info()["pass1"].split()[0]))
    if switched is None and r.jit_status()[0] == _capi.SSX_JIT_STATE_SPECIALISED: switched = time.time() - t0
    if switched is not None and time.time() - t0 > switched + 0.3: break
print("   " + " ".join("%.2fs:%.1f%s" % (t, ms, "" if k == "generic" else "*") for t, ms, k in trace[:6] + trace[-8:]), "  (* = the scene's own kernels)")
print("   switched after %.2f s of rendering (%d steps on the generic kernel)" % (switched or -1, sum(1 for x in trace if x[2] == "generic")))
jit = rate(r, "   steady state")
child = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--spp", str(a.spp), "--res", str(a.res)], capture_output=True, text=True, env=dict(os.environ))
print("4. second process (disk cache):", child.stdout.strip().splitlines()[-1] if child.stdout.strip() else child.stderr[-800:])
print("generic / built-in = %.3f   compiled / built-in = %.3f   (upload without / with a pending specialisation: %.3f / %.3f s)" % (gen / base, jit / base, up_generic, up))
