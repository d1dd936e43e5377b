"""Longer version of stress_parity.py: many seeds and shapes against the oracle (bit for bit)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
from simple_spectral_amd import Options, Renderer
def bits(a): return np.ascontiguousarray(a, np.float32).view(np.uint32)
bad = 0; total = 0
rng = np.random.default_rng(2026)
cases = []
N_CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for seed in range(1, N_CASES + 1):
    scene = ("cornell-srgb", "cornell", "plane-srgb")[seed % 3]
    W, H = int(rng.integers(40, 600)), int(rng.integers(40, 600))
    spp = int(rng.integers(3, 40))
    kw = dict(observer=2006) if seed % 4 == 0 else {}
    if seed % 3 == 1: kw["spp_per_launch"] = int(rng.integers(1, spp + 1))   # odd launch sizes: partial units and cohorts
    cases.append((scene, W, H, spp, seed, kw))
oracles = {}
for scene, W, H, spp, seed, kw in cases:
    tex = None if scene == "cornell" else "crystal-lizard-512.png"
    r = Renderer(Options(scene_name=scene, res=(W, H), spp=spp, seed=seed, texture=tex, **kw))
    r.render_start(); r.render_wait()
    key = (scene, kw.get("observer", 1931))
    if key not in oracles: oracles[key] = ol.Oracle(scene, texture=tex, observer=key[1])
    t = time.time()
    ref = oracles[key].render(W, H, spp, seed=seed, nthreads=min(16, os.cpu_count() or 1))
    d = int((bits(r.xyza) != bits(ref)).sum()); bad += d; total += W * H * spp
    print("%-13s %4dx%-4d spp %3d seed %2d %-18s differing floats: %d (oracle %.1f s)" % (scene, W, H, spp, seed, kw, d, time.time() - t), flush=True)
print("samples checked: %.1f M, differing floats: %d" % (total / 1e6, bad))
sys.exit(1 if bad else 0)
