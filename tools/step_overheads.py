#!/usr/bin/env python3
"""What a bench step costs beyond its kernels: K steps of the headline config timed (wall clock, fenced) as
  A  render only                      B  render + per-kernel timing events (ssx_set_timing)
  C  render + copy to pinned host memory on a second stream (double-buffered)     D  B + C (bench.py's timed region)
in the order A B C D A D C B A, so that drift of the box (clocks) shows as a difference between the A's.
    python tools/step_overheads.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from simple_spectral_amd import Options, Renderer

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
r = Renderer(Options(scene_name="cornell-srgb", res=(512, 512), spp=256, texture=os.path.join(ROOT, "data", "scenes", "test-img.png")))
outs = [torch.zeros((512, 512, 4), device="cuda") for _ in range(2)]
hosts = [torch.empty((512, 512, 4), pin_memory=True) for _ in range(2)]
stream = torch.cuda.current_stream()
copy_stream = torch.cuda.Stream()


def run(timing, copy):
    r.set_timing(timing)
    ready = [torch.cuda.Event() for _ in range(2)]
    copied = [None, None]
    torch.cuda.synchronize()
    t = time.perf_counter()
    for k in range(K):
        if copied[k & 1] is not None:
            stream.wait_event(copied[k & 1])
        r.render_device(outs[k & 1].data_ptr(), stream.cuda_stream)
        if copy:
            ready[k & 1].record(stream)
            copy_stream.wait_event(ready[k & 1])
            with torch.cuda.stream(copy_stream):
                hosts[k & 1].copy_(outs[k & 1], non_blocking=True)
                copied[k & 1] = torch.cuda.Event()
                copied[k & 1].record(copy_stream)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / K * 1e3
    st = {k: round(v / K, 3) for k, v in r.get_timing().items()} if timing else None
    r.set_timing(False)
    return ms, st


for _ in range(5):
    r.render_device(outs[0].data_ptr(), stream.cuda_stream)
torch.cuda.synchronize()
names = {"A": (False, False), "B": (True, False), "C": (False, True), "D": (True, True)}
for n in "ABCDADCBA":
    ms, st = run(*names[n])
    print("%s timing=%d copy=%d  %.3f ms/step  %s" % (n, names[n][0], names[n][1], ms, st or ""))
