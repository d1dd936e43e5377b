"""ctypes access to oracle/_ref/*.so -- the reference's OWN self-contained sources (rgb2spec.c,
Meng et al.'s spectrum_grid.h, lodepng.cpp), compiled in place from /root/reference by
`make -C oracle ref` (TEST INFRASTRUCTURE).  The built files travel to the GPU box; where they are
absent (a checkout without /root/reference) the tests that need them skip.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")


def _load(name):
    path = os.path.join(REF_DIR, name)
    if not os.path.exists(path):
        if os.path.isdir("/root/reference"):
            import subprocess
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True, capture_output=True)
    if not os.path.exists(path):
        return None
    return C.CDLL(path)


class RGB2SpecStruct(C.Structure):          # src/jakob-and-hanika-2019/rgb2spec.h:9-13
    _fields_ = [("res", C.c_uint32), ("scale", C.POINTER(C.c_float)), ("data", C.POINTER(C.c_float))]


_cache = {}


def rgb2spec():
    if "jh" not in _cache:
        lib = _load("libref_rgb2spec.so")
        if lib is not None:
            lib.rgb2spec_load.restype = C.POINTER(RGB2SpecStruct)
            lib.rgb2spec_load.argtypes = [C.c_char_p]
            lib.rgb2spec_free.argtypes = [C.POINTER(RGB2SpecStruct)]
            lib.rgb2spec_fetch.argtypes = [C.POINTER(RGB2SpecStruct), C.POINTER(C.c_float), C.POINTER(C.c_float)]
            lib.rgb2spec_eval_precise.restype = C.c_float
            lib.rgb2spec_eval_precise.argtypes = [C.POINTER(C.c_float), C.c_float]
        _cache["jh"] = lib
    return _cache["jh"]


def meng():
    if "meng" not in _cache:
        lib = _load("libref_meng.so")
        if lib is not None:
            lib.ref_meng_xyz_to_p.restype = C.c_float
            lib.ref_meng_xyz_to_p.argtypes = [C.c_float, C.POINTER(C.c_float)]
            lib.ref_meng_dims.argtypes = [C.POINTER(C.c_int)]
            lib.ref_meng_params.argtypes = [C.POINTER(C.c_float)]
            lib.ref_meng_cell.argtypes = [C.c_int, C.POINTER(C.c_int)]
            lib.ref_meng_point.argtypes = [C.c_int, C.POINTER(C.c_float)]
        _cache["meng"] = lib
    return _cache["meng"]


def lodepng():
    if "png" not in _cache:
        lib = _load("libref_lodepng.so")
        if lib is not None:
            for f in (lib.ref_png_decode_rgb8, lib.ref_png_decode_rgba8):
                f.restype = C.c_uint
                f.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_ubyte)), C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
            lib.ref_png_encode_rgba8.restype = C.c_uint
            lib.ref_png_encode_rgba8.argtypes = [C.c_char_p, C.c_void_p, C.c_uint, C.c_uint]
            lib.ref_png_free.argtypes = [C.c_void_p]
        _cache["png"] = lib
    return _cache["png"]


def lodepng_decode(path, rgba=False):
    """-> (error_code, HxWx{3,4} uint8 array or None), decoded by the reference's lodepng."""
    lib = lodepng()
    buf = C.POINTER(C.c_ubyte)()
    w, h = C.c_uint(0), C.c_uint(0)
    fn = lib.ref_png_decode_rgba8 if rgba else lib.ref_png_decode_rgb8
    err = fn(os.fsencode(path), C.byref(buf), C.byref(w), C.byref(h))
    if err:
        return err, None
    n = 4 if rgba else 3
    arr = np.ctypeslib.as_array(buf, shape=(h.value, w.value, n)).copy()
    lib.ref_png_free(buf)
    return 0, arr


def meng_table():
    """The Meng et al. grid as arrays read out of the reference's header (via libref_meng.so):
    dict(grid_w, grid_h, n_points, n_samples, sample_min, sample_max, xy_to_uv[6], cells[w*h,8] i32,
    points[n, 4+n_samples] f32)."""
    lib = meng()
    dims = (C.c_int * 4)()
    lib.ref_meng_dims(dims)
    gw, gh, npts, ns = (int(v) for v in dims)
    par = (C.c_float * 9)()
    lib.ref_meng_params(par)
    cells = np.zeros((gw * gh, 8), np.int32)
    for c in range(gw * gh):
        lib.ref_meng_cell(c, cells[c].ctypes.data_as(C.POINTER(C.c_int)))
    points = np.zeros((npts, 4 + ns), np.float32)
    for p in range(npts):
        lib.ref_meng_point(p, points[p].ctypes.data_as(C.POINTER(C.c_float)))
    return dict(grid_w=gw, grid_h=gh, n_points=npts, n_samples=ns, sample_min=float(par[0]), sample_max=float(par[1]),
                xy_to_uv=np.array(par[2:8], np.float32), equal_energy_reflectance=float(par[8]), cells=cells, points=points)
