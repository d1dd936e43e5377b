"""Rate of the host-buffer entry points (ssx_render_start + ssx_render_wait: the XYZA image crosses
PCIe to a host buffer) next to the device-resident rate bench.py reports."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simple_spectral_amd import Options, Renderer
for (W, spp) in ((512, 256), (2048, 16)):
    r = Renderer(Options(scene_name="cornell-srgb", res=(W, W), spp=spp, texture="crystal-lizard-512.png"))
    for _ in range(2):
        r.render_start(); r.render_wait()
    t = time.perf_counter()
    n = 5
    for _ in range(n):
        r.render_start(); r.render_wait()
    dt = (time.perf_counter() - t) / n
    print("cornell-srgb %dx%d spp=%d host path: %.2f ms  %.1f Msamples/s (XYZA image %d MiB over PCIe, incl. the host-side XYZ->sRGB)" % (W, W, spp, dt * 1e3, W * W * spp / dt / 1e6, W * W * 16 >> 20))
