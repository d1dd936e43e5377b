#!/usr/bin/env python3
"""A launch-bound render (BASELINE configs[0]: cornell-srgb 128 x 128 at 16 spp = 0.26 M samples: five kernels and three fills) enqueued call
by call against the same render captured once into a hipGraph and replayed (include/ssx.h: ssx_render_device on a capturing stream).
    python tools/graph_rate.py [--res 128 --spp 16]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from simple_spectral_amd import Options, Renderer
ap = argparse.ArgumentParser(); ap.add_argument("--res", type=int, default=128); ap.add_argument("--spp", type=int, default=16)
a = ap.parse_args()
r = Renderer(Options(scene_name="cornell-srgb", res=(a.res, a.res), spp=a.spp, texture="crystal-lizard-512.png"))
out = torch.zeros((a.res, a.res, 4), device="cuda")
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(3): r.render_device(out.data_ptr(), side.cuda_stream)
    side.synchronize(); r.render_device_wait()
    t = time.perf_counter()
    for _ in range(200): r.render_device(out.data_ptr(), side.cuda_stream)
    side.synchronize(); eager = (time.perf_counter() - t) / 200
    r.render_device_wait()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        r.render_device(out.data_ptr(), torch.cuda.current_stream().cuda_stream)
for _ in range(3): g.replay()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(200): g.replay()
torch.cuda.synchronize(); graph = (time.perf_counter() - t) / 200
n = a.res * a.res * a.spp
print("cornell-srgb %dx%d spp %d: call by call %.3f ms (%.0f Msamples/s), hipGraph replay %.3f ms (%.0f Msamples/s)" % (a.res, a.res, a.spp, eager * 1e3, n / eager / 1e6, graph * 1e3, n / graph / 1e6))
