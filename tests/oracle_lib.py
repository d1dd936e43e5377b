"""ctypes binding of the CPU oracle (oracle/libssx_oracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
DATA_DIR = os.path.join(ROOT, "data")


class V3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class V2(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class Rng(C.Structure):
    _fields_ = [("state", C.c_uint64), ("inc", C.c_uint64)]


class SphTri(C.Structure):
    _fields_ = [("A", V3), ("B", V3), ("C", V3)] + [
        (n, C.c_float)
        for n in (
            "a b c sin_a sin_b sin_c cos_a cos_b cos_c alpha beta gamma "
            "cos_alpha cos_beta cos_gamma surface_area"
        ).split()
    ]


class Vertex(C.Structure):
    _fields_ = [("pos", V3), ("st", V2)]


class Tri(C.Structure):
    _fields_ = [("verts", Vertex * 3), ("normal", V3)]


class Ray(C.Structure):
    _fields_ = [("orig", V3), ("dir", V3)]


class Hit(C.Structure):
    _fields_ = [("prim", C.c_int), ("normal", V3), ("st", V2), ("dist", C.c_float)]


class Stats(C.Structure):
    _fields_ = [
        (n, C.c_uint64)
        for n in "rays tri_tests tri_edge_pass tri_f64 interactions spectrum_lookups tex_samples".split()
    ] + [("path_len_hist", C.c_uint64 * 11), ("samples", C.c_uint64), ("hits", C.c_uint64)] + [
        (n, C.c_uint64)
        for n in ("sphtri_regular sphtri_half_pi sphtri_only_a sphtri_nan light_pdf_inf arvo_denom_zero "
                  "arvo_sin_alpha_le0 funcbar_zero coshemi_retries lemire_redraws nee_front nee_visible draws").split()
    ]

    def as_dict(self):
        return {n: (list(getattr(self, n)) if n == "path_len_hist" else int(getattr(self, n))) for n, _ in self._fields_}


class QuadIn(C.Structure):
    _fields_ = [("pos", (C.c_float * 3) * 4), ("st", (C.c_float * 2) * 4), ("material", C.c_int), ("kind", C.c_int)]


class MaterialIn(C.Structure):
    _fields_ = [("kind", C.c_int), ("albedo_mode", C.c_int), ("albedo_spectrum", C.c_int), ("texture", C.c_int),
                ("emission_spectrum", C.c_int)]


class SpectrumIn(C.Structure):
    _fields_ = [("n", C.c_int), ("low", C.c_float), ("high", C.c_float), ("data", C.POINTER(C.c_float))]


class TextureIn(C.Structure):
    _fields_ = [("w", C.c_int), ("h", C.c_int), ("rgb", C.POINTER(C.c_uint8))]


def build(force=False):
    """(Re)build the oracle shared objects with oracle/Makefile when gcc is available."""
    so = os.path.join(ORACLE_DIR, "libssx_oracle.so")
    if force or not os.path.exists(so) or _stale(so):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return so


def _stale(so):
    t = os.path.getmtime(so)
    deps = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
    deps.append(os.path.join(ROOT, "include", "ssx_fmath.h"))
    return any(os.path.getmtime(d) > t for d in deps)


_libs = {}


def load(variant=""):
    """variant '' = build-defined transcendentals (the parity oracle); 'libm' = glibc variant; 'refshape' = the oracle's arithmetic in the reference binary's call structure (bench.py: cpu_baseline.reference_equivalent)."""
    if variant in _libs:
        return _libs[variant]
    build()
    name = "libssx_oracle.so" if not variant else "libssx_oracle_%s.so" % variant
    lib = C.CDLL(os.path.join(ORACLE_DIR, name))
    vp, f32p, u8p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint8)
    lib.orc_last_error.restype = C.c_char_p
    lib.orc_color_create.restype = vp
    lib.orc_color_create.argtypes = [C.c_char_p, C.c_int]
    lib.orc_color_destroy.argtypes = [vp]
    lib.orc_color_set_jh.argtypes = [vp, C.c_int, vp, vp]
    lib.orc_jh_fetch.argtypes = [vp, f32p, f32p]
    lib.orc_jh_eval_precise.restype = C.c_float
    lib.orc_jh_eval_precise.argtypes = [f32p, C.c_float]
    lib.orc_color_set_rgb_mode.argtypes = [vp, C.c_int]
    lib.orc_color_set_meng.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, vp, vp, vp]
    lib.orc_meng_xyz_to_p.restype = C.c_float
    lib.orc_meng_xyz_to_p.argtypes = [vp, C.c_float, f32p]
    lib.orc_scene_create.restype = vp
    lib.orc_scene_create.argtypes = [vp, C.c_char_p, C.c_char_p, vp, C.c_int, C.c_int, C.c_float]
    lib.orc_scene_destroy.argtypes = [vp]
    lib.orc_scene_create_custom.restype = vp
    lib.orc_scene_create_custom.argtypes = [vp, C.POINTER(C.c_double), f32p, C.POINTER(SpectrumIn), C.c_int, C.POINTER(MaterialIn), C.c_int,
                                            C.POINTER(TextureIn), C.c_int, C.POINTER(QuadIn), C.c_int]
    lib.orc_scene_quad_normals.argtypes = [vp, C.c_int, f32p]
    lib.orc_scene_set_camera_dir.argtypes = [vp, f32p]
    lib.orc_scene_set_camera_dir.restype = None
    lib.orc_scene_light.argtypes = [vp, C.c_int]
    lib.orc_scene_set_material_kind.argtypes = [vp, C.c_int, C.c_int]
    lib.orc_scene_quad_material.argtypes = [vp, C.c_int]
    lib.orc_scene_material_rgb.argtypes = [vp, C.c_int, f32p]
    lib.orc_seed_sample.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(Rng)]
    lib.orc_render_sample.argtypes = [vp, vp, C.POINTER(Rng), C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                      C.c_int, f32p, C.POINTER(Stats)]
    lib.orc_render.restype = C.c_int
    lib.orc_render.argtypes = [vp, vp, C.c_uint64] + [C.c_size_t] * 7 + [C.c_int, C.c_int, vp, C.POINTER(Stats)]
    lib.orc_xyza_to_srgba.argtypes = [vp, vp, vp, C.c_size_t]
    lib.orc_debug_set_stats.argtypes = [C.POINTER(Stats)]
    lib.orc_rng_seed_u32.argtypes = [C.POINTER(Rng), C.c_uint32]
    lib.orc_rng_next.restype = C.c_uint32
    lib.orc_rng_next.argtypes = [C.POINTER(Rng)]
    lib.orc_rand_1f.restype = C.c_float
    lib.orc_rand_1f.argtypes = [C.POINTER(Rng)]
    lib.orc_rand_1d.restype = C.c_double
    lib.orc_rand_1d.argtypes = [C.POINTER(Rng)]
    lib.orc_rand_choice.restype = C.c_size_t
    lib.orc_rand_choice.argtypes = [C.POINTER(Rng), C.c_size_t]
    lib.orc_get_hashed_u32.restype = C.c_uint64
    lib.orc_get_hashed_u32.argtypes = [C.c_uint32]
    for fn in (lib.orc_sinf, lib.orc_cosf, lib.orc_acosf):
        fn.restype = C.c_float
        fn.argtypes = [C.c_float]
    lib.orc_spectrum_hero.argtypes = [vp, C.c_float, C.c_float, f32p]
    lib.orc_sphtri_make.argtypes = [V3, V3, V3, C.POINTER(SphTri)]
    lib.orc_rand_toward_sphericaltri.restype = V3
    lib.orc_rand_toward_sphericaltri.argtypes = [C.POINTER(Rng), C.POINTER(SphTri)]
    lib.orc_rand_coshemi.restype = V3
    lib.orc_rand_coshemi.argtypes = [C.POINTER(Rng), f32p]
    lib.orc_get_rotated_to.restype = V3
    lib.orc_get_rotated_to.argtypes = [V3, V3]
    lib.orc_tri_intersect.restype = C.c_int
    lib.orc_tri_intersect.argtypes = [C.POINTER(Tri), C.POINTER(Ray), C.POINTER(Hit), C.c_int, vp]
    lib.orc_scene_intersect.restype = C.c_int
    lib.orc_scene_intersect.argtypes = [vp, C.POINTER(Ray), C.POINTER(Hit), C.c_int, vp]
    lib.orc_lrgb_to_specrefl.argtypes = [vp, f32p, C.c_float, f32p]
    lib.orc_specradflux_to_ciexyz_hero.argtypes = [vp, f32p, C.c_float, f32p]
    lib.orc_srgb_to_lrgb.argtypes = [f32p, f32p]
    lib.orc_lrgb_to_srgb.argtypes = [f32p, f32p]
    lib.orc_round_trip_max_error.restype = C.c_float
    lib.orc_round_trip_max_error.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    lib.orc_round_trip_lrgb.argtypes = [vp, f32p, f32p]
    lib.orc_scene_pv_inv.restype = C.POINTER(C.c_double)
    lib.orc_scene_pv_inv.argtypes = [vp]
    lib.orc_scene_cam_pos.restype = f32p
    lib.orc_scene_cam_pos.argtypes = [vp]
    lib.orc_scene_counts.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.orc_color_spectrum.restype = vp
    lib.orc_color_spectrum.argtypes = [vp, C.c_char_p]
    lib.orc_color_matrix.restype = f32p
    lib.orc_color_matrix.argtypes = [vp, C.c_char_p]
    lib.orc_color_d65_rad_xyz.restype = f32p
    lib.orc_color_d65_rad_xyz.argtypes = [vp]
    lib.orc_spectrum_info.restype = C.c_int
    lib.orc_spectrum_info.argtypes = [vp, f32p, f32p, f32p, C.POINTER(f32p)]
    _libs[variant] = lib
    return lib


def load_texture(path):
    """Decode a PNG to RGB8 rows top-to-bottom (what lodepng::decode(..., LCT_RGB) yields)."""
    from PIL import Image

    im = Image.open(path).convert("RGB")
    a = np.ascontiguousarray(np.asarray(im, dtype=np.uint8))
    return a


class Oracle:
    """Colour tables + one scene, with render helpers.  observer: 1931 | 2006."""

    def __init__(self, scene="cornell-srgb", observer=1931, texture="test-img.png", light_scale=30.0,
                 variant="", data_dir=DATA_DIR, jh=None, meng=None, rgb=False, custom=None):
        """jh: (res, scale, data) Jakob-Hanika model -> RENDER_MODE_SPECTRAL_JH; meng: grid dict as
        returned by ref_lib.meng_table() / simple_spectral_amd.meng.load_table -> RENDER_MODE_SPECTRAL_MENG;
        neither -> "ours".  rgb=True -> RENDER_MODE_RGB (no spectra; renders return lRGB+A)."""
        self.lib = load(variant)
        self.color = self.lib.orc_color_create(data_dir.encode(), observer)
        if not self.color:
            raise RuntimeError(self.lib.orc_last_error().decode())
        if jh is not None:
            res, scale, data = jh
            scale = np.ascontiguousarray(scale, dtype=np.float32); data = np.ascontiguousarray(data, dtype=np.float32)
            if self.lib.orc_color_set_jh(self.color, int(res), scale.ctypes.data, data.ctypes.data) != 0:
                raise RuntimeError(self.lib.orc_last_error().decode())
        if rgb:
            self.lib.orc_color_set_rgb_mode(self.color, 1)
        if meng is not None:
            cells = np.ascontiguousarray(meng["cells"], dtype=np.int32); points = np.ascontiguousarray(meng["points"], dtype=np.float32)
            m = np.ascontiguousarray(meng["xy_to_uv"], dtype=np.float32)
            if self.lib.orc_color_set_meng(self.color, meng["grid_w"], meng["grid_h"], meng["n_points"], meng["n_samples"],
                                           C.c_float(meng["sample_min"]), C.c_float(meng["sample_max"]), m.ctypes.data,
                                           cells.ctypes.data, points.ctypes.data) != 0:
                raise RuntimeError(self.lib.orc_last_error().decode())
        if custom is not None:
            # custom(lib, color) -> scene handle (tests/custom_scene.py: orc_scene_create_custom)
            self.texture = None
            self.scene = custom(self.lib, self.color)
            if not self.scene:
                raise RuntimeError(self.lib.orc_last_error().decode())
            self.scene_name = "custom"
            return
        tex = None
        if texture is not None and scene != "cornell":
            tex = texture if isinstance(texture, np.ndarray) else load_texture(
                texture if os.path.isabs(texture) else os.path.join(data_dir, "scenes", texture))
        self.texture = tex
        tp, tw, th = (tex.ctypes.data, tex.shape[1], tex.shape[0]) if tex is not None else (None, 0, 0)
        self.scene = self.lib.orc_scene_create(self.color, scene.encode(), data_dir.encode(), tp, tw, th,
                                               C.c_float(light_scale))
        if not self.scene:
            raise RuntimeError(self.lib.orc_last_error().decode())
        self.scene_name = scene

    def close(self):
        if self.scene:
            self.lib.orc_scene_destroy(self.scene)
            self.scene = None
        if self.color:
            self.lib.orc_color_destroy(self.color)
            self.color = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, W, H, spp, seed=0, rect=None, indirect_only=False, nthreads=0, stats=False, els=True, flat_field=True):
        out = np.zeros((H, W, 4), dtype=np.float32)
        i0, j0, i1, j1 = rect if rect else (0, 0, W, H)
        st = Stats() if stats else None
        rc = self.lib.orc_render(self.color, self.scene, seed, W, H, i0, j0, i1, j1, spp, int(indirect_only) | (0 if els else 2) | (0 if flat_field else 4),
                                 nthreads, out.ctypes.data, C.byref(st) if stats else None)
        assert rc == 0
        return (out, st) if stats else out

    def sample(self, i, j, k, W, H, seed=0, indirect_only=False):
        rng = Rng()
        self.lib.orc_seed_sample(seed, j * W + i, k, C.byref(rng))
        out = (C.c_float * 4)()
        self.lib.orc_render_sample(self.color, self.scene, C.byref(rng), i, j, W, H, int(indirect_only), out, None)
        return np.array(out[:], dtype=np.float32)

    def samples(self, W, H, spp, seed=0, rect=None, indirect_only=False, els=True, flat_field=True):
        """Per-sample results for the pixel rectangle: (xyza [h, w, spp, 4] float32, final PCG32 state
        [h, w, spp] uint64 -- i.e. the draws consumed -- and the summed stats)."""
        i0, j0, i1, j1 = rect if rect else (0, 0, W, H)
        xyza = np.zeros((j1 - j0, i1 - i0, spp, 4), dtype=np.float32)
        state = np.zeros((j1 - j0, i1 - i0, spp), dtype=np.uint64)
        st = Stats()
        rng = Rng()
        out = (C.c_float * 4)()
        flags = int(indirect_only) | (0 if els else 2) | (0 if flat_field else 4)
        for j in range(j0, j1):
            for i in range(i0, i1):
                for k in range(spp):
                    self.lib.orc_seed_sample(seed, j * W + i, k, C.byref(rng))
                    self.lib.orc_render_sample(self.color, self.scene, C.byref(rng), i, j, W, H, flags, out, C.byref(st))
                    xyza[j - j0, i - i0, k] = out[:]
                    state[j - j0, i - i0, k] = rng.state
        return xyza, state, st

    def to_srgba(self, xyza):
        xyza = np.ascontiguousarray(xyza, dtype=np.float32)
        out = np.empty_like(xyza)
        self.lib.orc_xyza_to_srgba(self.color, xyza.ctypes.data, out.ctypes.data, xyza.size // 4)
        return out

    def spectrum(self, name):
        sp = self.lib.orc_color_spectrum(self.color, name.encode())
        low, high, dr = C.c_float(), C.c_float(), C.c_float()
        data = C.POINTER(C.c_float)()
        n = self.lib.orc_spectrum_info(sp, C.byref(low), C.byref(high), C.byref(dr), C.byref(data))
        return np.ctypeslib.as_array(data, shape=(n,)).copy(), low.value, high.value, dr.value
