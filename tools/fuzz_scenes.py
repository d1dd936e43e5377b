#!/usr/bin/env python3
"""Differential fuzz on the GPU box: N random scenes (tests/crafted.py random_scene: primitives, materials, lights, camera and
render switches drawn from the seed), each rendered by the library and by the CPU oracle, images compared bit for bit.
The GPU suite runs the first 48 seeds per sample; this is the long version.     usage: python tools/fuzz_scenes.py [--warped] [first [count]]
--warped: tests/crafted.py warped_builtin instead -- the built-in scenes with every distinct corner moved, on the topology-specialised kernels.
(test tooling: uses the oracle as the checker)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import crafted  # noqa: E402
from simple_spectral_amd.renderer import Options, Renderer  # noqa: E402


def main():
    warped = "--warped" in sys.argv       # the built-in scenes with every corner moved (tests/crafted.py warped_builtin): the topology-specialised kernels
    if warped:
        sys.argv.remove("--warped")
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    bad, t0, prims, nan_px, samples = [], time.time(), 0, 0, 0
    for seed in range(first, first + count):
        if warped:
            c, base, o = crafted.warped_builtin(seed)
            orc = c.oracle()
            r = Renderer(Options(scene_name=base, res=(8, 8), spp=1, texture=None if base == "cornell" else "test-img.png", observer=c.observer))
            r.upload_scene_desc(c.desc(orc))
            assert r.plan_info()["pass1"] == ("plane topology" if base == "plane-srgb" else "cornell topology")
        else:
            c, o = crafted.random_scene(seed)
            orc = c.oracle()
            r = Renderer(Options(scene_name="cornell", res=(8, 8), spp=1, observer=c.observer))
            r.upload_scene_desc(c.desc(orc))
        g = np.random.default_rng(77 + seed)                                   # image shape, samples and launch chunking vary too
        W, H, spp = int(g.integers(1, 71)), int(g.integers(1, 61)), int(g.integers(1, 10))
        r.options.spp_per_launch = int(g.integers(0, spp + 1))                  # 0: the library's choice
        r.options.tile_major = bool(g.integers(0, 2))                           # the order of the work: through the samples / through the tiles
        r.options.tile_skew = int(g.integers(0, 4))                             # ... and of the tile list (one device: the order of the units only)
        samples += W * H * spp
        r.options.res = (W, H); r.options.spp = spp; r.options.seed = seed
        r.options.indirect_only = o["indirect_only"]; r.options.explicit_light_sampling = o["els"]; r.options.flat_field_correction = o["flat_field"]
        r.xyza = np.zeros((H, W, 4), dtype=np.float32)
        r.render_start(); r.render_wait()
        ref = orc.render(W, H, spp, seed=seed, indirect_only=o["indirect_only"], els=o["els"], flat_field=o["flat_field"])
        g, e = r.xyza.view(np.uint32), ref.view(np.uint32)
        same = (g == e) | (np.isnan(r.xyza) & np.isnan(ref))
        prims += len(c.quads); nan_px += int(np.isnan(ref).any(axis=-1).sum())
        if not same.all():
            bad.append((seed, int((~same).sum())))
            print("MISMATCH seed", seed, "floats", int((~same).sum()), o, flush=True)
        r.close()
    print(("warped built-in scenes (cornell-srgb / plane-srgb / cornell on their topology kernels): " if warped else "") + "scenes %d (seeds %d..%d), %d primitives, %d samples (images of 1..70 x 1..60 pixels, 1..9 spp, random launch chunking); pixels with a NaN in the oracle's image (both sides agree): %d; mismatching scenes: %d; %.0f s"
          % (count, first, first + count - 1, prims, samples, nan_px, len(bad), time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
