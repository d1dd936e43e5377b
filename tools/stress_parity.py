"""Stress check for rare-event bugs (memory ordering in the shadow-ray queue / fold, unit rotation):
big renders against the oracle bit for bit, and run-to-run determinism at sizes the oracle cannot reach."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import oracle_lib as ol
from simple_spectral_amd import Options, Renderer

def bits(a): return np.ascontiguousarray(a, np.float32).view(np.uint32)
bad = 0
for scene, W, H, spp, seed, kw in (("cornell-srgb", 512, 512, 32, 11, {}), ("cornell", 384, 256, 48, 12, {}), ("plane-srgb", 512, 512, 32, 13, {}),
                                   ("cornell-srgb", 333, 217, 40, 14, dict(observer=2006)), ("cornell-srgb", 256, 256, 64, 15, dict(explicit_light_sampling=False)),
                                   ("cornell-srgb", 512, 512, 24, 16, dict(indirect_only=True))):
    tex = None if scene == "cornell" else "crystal-lizard-512.png"
    r = Renderer(Options(scene_name=scene, res=(W, H), spp=spp, seed=seed, texture=tex, **kw))
    r.render_start(); r.render_wait()
    t = time.time()
    o = ol.Oracle(scene, texture=tex, observer=kw.get("observer", 1931))
    if scene == "plane-srgb" and not kw.get("explicit_light_sampling", True):
        o.lib.orc_scene_set_material_kind(o.scene, o.lib.orc_scene_quad_material(o.scene, 0), 1)
    ref = o.render(W, H, spp, seed=seed, indirect_only=kw.get("indirect_only", False), els=kw.get("explicit_light_sampling", True))
    d = int((bits(r.xyza) != bits(ref)).sum())
    bad += d
    print("%-13s %4dx%-4d spp %3d %-34s differing floats: %d   (oracle %.1f s)" % (scene, W, H, spp, kw, d, time.time() - t), flush=True)
for scene, W, spp in (("cornell-srgb", 2048, 48), ("plane-srgb", 2048, 256), ("cornell", 1024, 384)):
    tex = None if scene == "cornell" else "crystal-lizard-512.png"
    r = Renderer(Options(scene_name=scene, res=(W, W), spp=spp, seed=5, texture=tex))
    outs = []
    for rep in range(3):
        out = torch.zeros((W, W, 4), device="cuda")
        r.render_device(out.data_ptr(), torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
    d = int((bits(outs[0]) != bits(outs[1])).sum() + (bits(outs[0]) != bits(outs[2])).sum())
    bad += d
    print("%-13s %4d^2 spp %3d three runs: differing floats %d, finite %s" % (scene, W, spp, d, bool(np.isfinite(outs[0]).all())), flush=True)
print("TOTAL differing:", bad)
sys.exit(1 if bad else 0)
