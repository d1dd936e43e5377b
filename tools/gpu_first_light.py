"""First-light check on the GPU box: HIP megakernel vs the CPU oracle, bit for bit."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
from simple_spectral_amd import Options, Renderer

def compare(scene, W, H, spp, tex="test-img.png", observer=1931):
    o = ol.Oracle(scene, texture=tex if scene != "cornell" else None, observer=observer)
    t = time.time(); ref = o.render(W, H, spp); tc = time.time() - t
    r = Renderer(Options(scene_name=scene, res=(W, H), spp=spp, texture=tex, observer=observer))
    t = time.time(); r.render_start(); r.render_wait(); tg = time.time() - t
    got = r.xyza
    neq = (got.view(np.uint32) != ref.view(np.uint32))
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3)
    print("%-13s %dx%d spp=%d obs=%d: bitwise-different floats %d / %d, pixels %d, max rel %.3e | cpu %.2fs gpu %.3fs" % (
        scene, W, H, spp, observer, neq.sum(), neq.size, neq.any(axis=2).sum(), rel.max(), tc, tg), flush=True)
    if neq.any():
        idx = np.argwhere(neq.any(axis=2))[:5]
        for j, i in idx:
            print("   pixel", i, j, got[j, i], ref[j, i])
    print("   kernel:", r.kernel_info())
    return neq.sum()

bad = 0
bad += compare("cornell", 64, 64, 4)
bad += compare("cornell-srgb", 64, 64, 4)
bad += compare("plane-srgb", 64, 64, 4)
bad += compare("cornell-srgb", 128, 128, 16)
bad += compare("cornell-srgb", 64, 64, 4, observer=2006)
bad += compare("cornell-srgb", 50, 37, 3)
# timing
r = Renderer(Options(scene_name="cornell-srgb", res=(512, 512), spp=64, texture="crystal-lizard-512.png"))
for rep in range(2):
    t = time.time(); r.render_start(); r.render_wait(); dt = time.time() - t
    print("cornell-srgb 512x512 spp=64: %.3f s -> %.1f Msamples/s" % (dt, 512 * 512 * 64 / dt / 1e6), flush=True)
print("TOTAL bitwise mismatches:", bad)
