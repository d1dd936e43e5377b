"""Generate tests/golden/*.npz from the CPU oracle (oracle/libssx_oracle.so) in this container.

The reference holds no vectors for the integrator and cannot be built here,
so these are regression vectors of the oracle itself: they freeze its behaviour so that (a) an
accidental change of the oracle is caught on CPU and (b) the GPU box, where gcc output could in
principle differ, checks the HIP path against numbers produced HERE.  Inputs are the seeding
contract's (seed, pixel, k) triples; outputs are XYZA floats.

    python tests/golden/make_goldens.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402

CASES = [
    # name, scene, observer, texture, W, H, spp, seed, indirect_only
    ("cornell_1931", "cornell", 1931, None, 32, 32, 8, 0, False),
    ("cornell_srgb_1931", "cornell-srgb", 1931, "test-img.png", 32, 32, 8, 0, False),
    ("plane_srgb_1931", "plane-srgb", 1931, "test-img.png", 32, 32, 8, 0, False),
    ("cornell_srgb_2006", "cornell-srgb", 2006, "test-img.png", 32, 32, 8, 7, False),
    ("cornell_srgb_indirect", "cornell-srgb", 1931, "test-img.png", 24, 16, 4, 3, True),
    ("cornell_srgb_ragged", "cornell-srgb", 1931, "test-img.png", 21, 13, 5, 11, False),
]


def main():
    out = {}
    for name, scene, obs, tex, W, H, spp, seed, io in CASES:
        o = ol.Oracle(scene, observer=obs, texture=tex)
        img = o.render(W, H, spp, seed=seed, indirect_only=io, nthreads=1)
        out[name + "__image"] = img
        out[name + "__meta"] = np.array([W, H, spp, seed, int(io), obs], dtype=np.int64)
        # per-sample vectors for a few pixels
        rs = np.random.RandomState(1234)
        ijk = np.stack([rs.randint(0, W, 48), rs.randint(0, H, 48), rs.randint(0, 1000, 48)], axis=1).astype(np.int64)
        samples = np.stack([o.sample(int(i), int(j), int(k), W, H, seed=seed, indirect_only=io) for i, j, k in ijk])
        out[name + "__ijk"] = ijk
        out[name + "__samples"] = samples
        o.close()
    np.savez_compressed(os.path.join(HERE, "integrator_goldens.npz"), **out)
    print("wrote integrator_goldens.npz:", sum(v.nbytes for v in out.values()), "bytes raw")


if __name__ == "__main__":
    main()
