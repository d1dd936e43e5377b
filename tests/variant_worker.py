"""Worker of tests/test_gpu_variants.py: ONE process = ONE kernel variant (the library / environment switches are read when the
library is loaded), runs the hand-over stress cases and a slice of the differential scene fuzz against the CPU oracle, prints one
JSON line.  Test infrastructure (uses the oracle as the checker).      usage: python tests/variant_worker.py <first seed> <count>"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
import crafted  # noqa: E402
import oracle_lib as ol  # noqa: E402
from simple_spectral_amd import Options, Renderer, _capi  # noqa: E402
from simple_spectral_amd import dist as sdist  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def stress(report):
    """The cases that exercise the inter-wave hand-over of the ordered pixel sums (ssx_kernels.hip unit_fold / sums_chain) hardest."""
    import torch
    bad = []
    # (a) one tile, hundreds of its units in flight at once; split over launches; a ragged tile
    for (W, H, spp, chunk) in ((16, 8, 2048, 0), (8, 8, 1536, 500), (11, 5, 777, 0)):
        r = Renderer(Options(scene_name="cornell-srgb", res=(W, H), spp=spp, seed=33, texture="test-img.png", spp_per_launch=chunk))
        r.render_start(); r.render_wait()
        ref = ol.Oracle("cornell-srgb", texture="test-img.png").render(W, H, spp, seed=33)
        if not np.array_equal(bits(r.xyza), bits(ref)):
            bad.append("chain %dx%d spp %d" % (W, H, spp))
        report.setdefault("units_parked", []).append(r.sums_info()["units_parked"])
    # (b) one rank's share of an 8-GPU render in miniature: every eighth tile at 8 x the samples per pixel (three quarters of the units park)
    r = Renderer(Options(scene_name="cornell-srgb", res=(64, 64), spp=512, seed=7, texture="test-img.png", tile_first=3, tile_stride=8, tile_skew=1))
    ref = ol.Oracle("cornell-srgb", texture="test-img.png").render(64, 64, 512, seed=7)
    mask = sdist.tile_owner_mask(64, 64, 3, 8, 1)
    ref[~mask] = 0.0
    out = torch.zeros((64, 64, 4), device="cuda")
    p0 = r.sums_info()["units_parked"]
    for _ in range(3):
        r.render_device(out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        if not np.array_equal(bits(out.cpu().numpy()), bits(ref)):
            bad.append("n8 share")
    units = 8 * 128 * 3
    report["n8_share_parked_frac"] = round((r.sums_info()["units_parked"] - p0) / units, 3)
    # (c) units that finish neck and neck (one sample per pixel each): every hand-over while the predecessor's stores are still on their way
    r = Renderer(Options(scene_name="cornell-srgb", res=(128, 128), spp=16, seed=2, texture="test-img.png"))
    ref = ol.Oracle("cornell-srgb", texture="test-img.png").render(128, 128, 16, seed=2)
    out = torch.zeros((128, 128, 4), device="cuda")
    for _ in range(20):
        r.render_device(out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        if not np.array_equal(bits(out.cpu().numpy()), bits(ref)):
            bad.append("neck and neck")
            break
    # (d) ~10 units per wave, lizard texture
    r = Renderer(Options(scene_name="cornell-srgb", res=(256, 256), spp=32, seed=11, texture="crystal-lizard-512.png"))
    r.render_start(); r.render_wait()
    report["kernel"] = r.plan_info()["kernel"]
    report["queue_words"] = None
    ref = ol.Oracle("cornell-srgb", texture="crystal-lizard-512.png").render(256, 256, 32, seed=11)
    if not np.array_equal(bits(r.xyza), bits(ref)):
        bad.append("many units per wave")
    return bad


def fuzz(first, count, report):
    bad, samples = [], 0
    for seed in range(first, first + count):
        c, o = crafted.random_scene(seed)
        orc = c.oracle()
        r = Renderer(Options(scene_name="cornell", res=(8, 8), spp=1, observer=c.observer))
        r.upload_scene_desc(c.desc(orc))
        g = np.random.default_rng(77 + seed)
        W, H, spp = int(g.integers(1, 71)), int(g.integers(1, 61)), int(g.integers(1, 10))
        r.options.spp_per_launch = int(g.integers(0, spp + 1))
        r.options.tile_major = bool(g.integers(0, 2))
        r.options.tile_skew = int(g.integers(0, 4))
        samples += W * H * spp
        r.options.res = (W, H); r.options.spp = spp; r.options.seed = seed
        r.options.indirect_only = o["indirect_only"]; r.options.explicit_light_sampling = o["els"]; r.options.flat_field_correction = o["flat_field"]
        r.xyza = np.zeros((H, W, 4), dtype=np.float32)
        r.render_start(); r.render_wait()
        ref = orc.render(W, H, spp, seed=seed, indirect_only=o["indirect_only"], els=o["els"], flat_field=o["flat_field"])
        same = (r.xyza.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(r.xyza) & np.isnan(ref))
        if not same.all():
            bad.append(seed)
        r.close()
    report["fuzz_scenes"] = count
    report["fuzz_samples"] = samples
    return bad


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    t = time.time()
    report = {"library": os.path.basename(_capi.hip_lib()._name), "env": {k: v for k, v in os.environ.items() if k.startswith("SSX_")}}
    report["stress_failures"] = stress(report)
    report["stress_seconds"] = round(time.time() - t, 1)
    t = time.time()
    report["fuzz_mismatching_seeds"] = fuzz(first, count, report)
    report["fuzz_seconds"] = round(time.time() - t, 1)
    print(json.dumps(report), flush=True)
    return 1 if report["stress_failures"] or report["fuzz_mismatching_seeds"] else 0


if __name__ == "__main__":
    sys.exit(main())
