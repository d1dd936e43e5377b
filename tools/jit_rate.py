#!/usr/bin/env python3
"""Rates of the three pass-1 variants side by side (GPU box): the built-in Cornell topology, the same box with one corner
moved apart from its twin under the generic loop, and that scene with pass 1 compiled for its own topology at upload
(ssx_set_jit).  VERDICT r02 item 7: the last one within 3 % of the first.      python tools/jit_rate.py [--spp 256]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import custom_scene as cs
from simple_spectral_amd import Options, Renderer
ap = argparse.ArgumentParser(); ap.add_argument("--spp", type=int, default=256); ap.add_argument("--res", type=int, default=512)
a = ap.parse_args()
W = H = a.res
def rate(r, label):
    out = torch.zeros((H, W, 4), device="cuda"); s = torch.cuda.current_stream()
    for _ in range(3): r.render_device(out.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): r.render_device(out.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print("%-46s %-36s %8.1f Msamples/s  %.3f ms" % (label, r.plan_info()["pass1"], W * H * a.spp / dt / 1e6, dt * 1e3), flush=True)
    return W * H * a.spp / dt
o = dict(scene_name="cornell-srgb", res=(W, H), spp=a.spp, texture="crystal-lizard-512.png")
base = rate(Renderer(Options(**o)), "cornell-srgb (built-in topology)")
c = cs.CustomScene("cornell-srgb", texture="crystal-lizard-512.png")
pos, st, m = c.quads[0]; pos = pos.copy(); pos[0, 0] += 1.0; c.quads[0] = (pos, st, m)
orc = c.oracle()
r = Renderer(Options(**o)); r.upload_scene_desc(c.desc(orc)); gen = rate(r, "one corner moved: generic loop")
r = Renderer(Options(jit_pass1=True, **o)); t = time.time(); r.upload_scene_desc(c.desc(orc)); up = time.time() - t
jit = rate(r, "one corner moved: compiled at upload (%.1f s)" % up)
print("generic / built-in = %.3f   compiled / built-in = %.3f" % (gen / base, jit / base))
