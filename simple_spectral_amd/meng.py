"""Meng et al. 2015 grid as a data file ("SSXMENG1", layout in host/meng2015.hpp).

The reference compiles the authors' tables in from a C header it vendors
(src/meng-et-al.-2015/spectra_xyz_5nm_380_780_0.97.h).  This package ships no copy of them:
convert your copy of that header once,

    python -m simple_spectral_amd.meng /path/to/spectra_xyz_5nm_380_780_0.97.h data/meng-et-al-2015-grid.bin

and pass the result as `meng_grid_path` / `--meng-grid`.  The converter reads the header as TEXT
(array initialisers), it does not compile or execute it.
"""
import re
import struct
import sys

import numpy as np

MAGIC = b"SSXMENG1"


def save_table(path, t):
    cells = np.ascontiguousarray(t["cells"], dtype="<i4").reshape(-1, 8)
    points = np.ascontiguousarray(t["points"], dtype="<f4").reshape(t["n_points"], 4 + t["n_samples"])
    assert cells.shape[0] == t["grid_w"] * t["grid_h"]
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<4I", t["grid_w"], t["grid_h"], t["n_points"], t["n_samples"]))
        f.write(struct.pack("<2f", t["sample_min"], t["sample_max"]))
        f.write(np.ascontiguousarray(t["xy_to_uv"], dtype="<f4").tobytes())
        f.write(cells.tobytes())
        f.write(points.tobytes())


def load_table(path):
    with open(path, "rb") as f:
        blob = f.read()
    if blob[:8] != MAGIC:
        raise ValueError("%s: not a Meng grid file" % path)
    gw, gh, npts, ns = struct.unpack_from("<4I", blob, 8)
    smin, smax = struct.unpack_from("<2f", blob, 24)
    m = np.frombuffer(blob, "<f4", 6, 32).copy()
    off = 56
    cells = np.frombuffer(blob, "<i4", gw * gh * 8, off).reshape(-1, 8).copy()
    off += cells.nbytes
    points = np.frombuffer(blob, "<f4", npts * (4 + ns), off).reshape(npts, 4 + ns).copy()
    return dict(grid_w=gw, grid_h=gh, n_points=npts, n_samples=ns, sample_min=smin, sample_max=smax,
                xy_to_uv=m, cells=cells, points=points)


_NUM = r"[-+]?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?)f?"


def _scalar(text, name):
    m = re.search(r"\b%s\s*=\s*(%s)\s*;" % (re.escape(name), _NUM), text)
    if not m:
        raise ValueError("header: `%s` not found" % name)
    return float(m.group(1).rstrip("f"))


def _array_body(text, name):
    m = re.search(r"\b%s\s*\[\s*\]\s*=\s*\{" % re.escape(name), text)
    if not m:
        raise ValueError("header: array `%s` not found" % name)
    depth, i = 1, m.end()
    while depth:
        c = text[i]
        depth += (c == "{") - (c == "}")
        i += 1
    return text[m.end():i - 1]


def table_from_header(path):
    """Parse the authors' header (text) into the table dict."""
    text = open(path, "r", encoding="utf-8", errors="replace").read()
    text = re.sub(r"//[^\n]*|/\*.*?\*/", "", text, flags=re.S)
    gw, gh = int(_scalar(text, "spectrum_grid_width")), int(_scalar(text, "spectrum_grid_height"))
    ns = int(_scalar(text, "spectrum_num_samples"))
    smin, smax = _scalar(text, "spectrum_sample_min"), _scalar(text, "spectrum_sample_max")
    nums = lambda body: [float(v.rstrip("f")) for v in re.findall(_NUM, body)]
    m = np.array(nums(_array_body(text, "spectrum_mat_xy_to_uv")), np.float32)   # float literals: double -> float, as the compiler does
    cells = np.array(nums(_array_body(text, "spectrum_grid")), np.float64).astype(np.int32).reshape(-1, 8)
    points = np.array(nums(_array_body(text, "spectrum_data_points")), np.float64).astype(np.float32).reshape(-1, 4 + ns)
    if m.size != 6 or cells.shape[0] != gw * gh:
        raise ValueError("header: unexpected table sizes")
    return dict(grid_w=gw, grid_h=gh, n_points=points.shape[0], n_samples=ns, sample_min=smin, sample_max=smax,
                xy_to_uv=m, cells=cells, points=points)


if __name__ == "__main__":
    if len(sys.argv) != 3:
        sys.exit(__doc__)
    t = table_from_header(sys.argv[1])
    save_table(sys.argv[2], t)
    print("wrote %s: %dx%d cells, %d points x %d samples" % (sys.argv[2], t["grid_w"], t["grid_h"], t["n_points"], t["n_samples"]))
