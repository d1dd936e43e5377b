"""Host-side mirror of the reference's Renderer (src/renderer.hpp:14-82) over the C ABI.

    opts = Options(scene_name="cornell-srgb", res=(512, 512), spp=256)
    r = Renderer(opts)            # builds tables + scene on the host, uploads to the GPU
    r.render_start(); r.render_wait()
    r.framebuffer                 # sRGB+A float32 [H, W, 4], row 0 = bottom (src/framebuffer.hpp:26-34)
    r.xyza                        # the XYZ+alpha means the kernel produced (parity metric)

Everything numeric happens in libssx_hip.so / libssx_host.so; there is no Python or PyTorch
implementation of the integrator to fall back to.
"""
import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Optional, Tuple

import numpy as np

from . import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_DATA_DIR = os.path.join(ROOT, "data")


class SsxError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("ssx error %d: %s" % (code, message))
        self.code = code


@dataclass
class Options:
    """Renderer::Options (src/renderer.hpp:16-29) plus the additive options of this build."""
    scene_name: str = "cornell-srgb"
    res: Tuple[int, int] = (512, 512)
    spp: int = 16
    indirect_only: bool = False
    output_path: str = ""
    # additive (defaults = the reference's compile-time defaults)
    observer: int = 1931                 # CIE_OBSERVER
    seed: int = 0
    texture: Optional[str] = None        # PNG path; default per scene below
    light_scale: float = 30.0            # lightsc (src/scene.cpp:291-293)
    explicit_light_sampling: bool = True  # EXPLICIT_LIGHT_SAMPLING (src/stdafx.hpp:44); False also makes
    #                                       plane-srgb's textured quad a mirror (src/scene.cpp:346-355)
    flat_field_correction: bool = True   # FLAT_FIELD_CORRECTION (src/stdafx.hpp:55); False: flux = radiance * dot(ray dir, camera.dir)
    render_mode: str = "spectral"        # "spectral" (RENDER_MODE_SPECTRAL) | "rgb" (RENDER_MODE_RGB, src/stdafx.hpp:91-93:
    #                                       no spectra; `xyza` then holds linear RGB + alpha, `uplift`/`observer` are unused)
    uplift: str = "ours"                 # RENDER_MODE_SPECTRAL_ALGNUM: "ours" (1) | "meng" (2, Meng et al. 2015) | "jh" (3, Jakob-Hanika 2019)
    meng_grid_path: Optional[str] = None  # "SSXMENG1" file converted from the authors' header (simple_spectral_amd/meng.py)
    jh_res: int = 64                     # resolution of the fitted JH model when no coefficient file exists
    jh_coeff_path: Optional[str] = None  # data/jakob-and-hanika-2019-srgb.coeff in the reference (missing blob)
    jit_pass1: Optional[bool] = None     # ssx_set_jit: kernels compiled for the mesh topology of a scene that matches no built-in one (hipRTC).
    #                                       None: in the background, when the scene has rendered enough (the library's default);
    #                                       True: at upload, on the calling thread; False: never (generic kernel)
    device: int = 0
    tile_first: int = 0
    tile_stride: int = 1
    tile_skew: int = 0                   # ssx_render_params.tile_skew: rotate tile row ty of the shared-out list by ty * tile_skew columns (diagonal instead of vertical stripes)
    spp_per_launch: int = 0
    tile_major: bool = False             # ssx_render_params.tile_major: walk through the tiles like the reference (a stopped render keeps
    #                                       finished tiles at full sample count, the rest untouched) instead of through the samples
    data_dir: str = field(default=DEFAULT_DATA_DIR)


def default_texture(data_dir):
    """The reference opens data/scenes/crystal-lizard-4096.png (src/scene.cpp:292,357), a blob
    missing from the repository; like the reference's own commented alternatives we fall back to
    the 512^2 version when it is absent."""
    for name in ("crystal-lizard-4096.png", "crystal-lizard-512.png", "test-img.png"):
        p = os.path.join(data_dir, "scenes", name)
        if os.path.exists(p):
            return p
    return None


class Scene:
    """Host-prepared scene + colour tables (libssx_host.so)."""

    def __init__(self, name, observer=1931, texture=None, light_scale=30.0, data_dir=DEFAULT_DATA_DIR,
                 uplift="ours", jh_res=64, jh_coeff_path=None, explicit_light_sampling=True, meng_grid_path=None, render_mode="spectral"):
        lib = _capi.host_lib()
        self._lib = lib
        self._h = C.c_void_p()
        tex_path = None
        self._tex = None
        if name != "cornell":
            if isinstance(texture, np.ndarray):
                self._tex = np.ascontiguousarray(texture, dtype=np.uint8)
            else:
                tex_path = texture or default_texture(data_dir)
                if tex_path and not tex_path.startswith("procedural:") and not os.path.isabs(tex_path) and not os.path.exists(tex_path):
                    tex_path = os.path.join(data_dir, "scenes", tex_path)
        tp, tw, th = (self._tex.ctypes.data, self._tex.shape[1], self._tex.shape[0]) if self._tex is not None else (None, 0, 0)
        if uplift not in ("ours", "meng", "jh"):
            raise SsxError(_capi.SSX_ERR_SCENE, "unsupported uplift %r (ours | meng | jh)" % (uplift,))
        if render_mode not in ("spectral", "rgb"):
            raise SsxError(_capi.SSX_ERR_SCENE, "unsupported render mode %r (spectral | rgb)" % (render_mode,))
        code = {"ours": _capi.SSX_UPLIFT_OURS, "meng": _capi.SSX_UPLIFT_MENG, "jh": _capi.SSX_UPLIFT_JH}[uplift]
        table_path = (meng_grid_path or os.path.join(data_dir, "meng-et-al-2015-grid.bin")) if uplift == "meng" else jh_coeff_path
        rc = lib.ssh_scene_create_ex(name.encode(), data_dir.encode(), observer, tp, tw, th,
                                     tex_path.encode() if tex_path else None, C.c_float(light_scale),
                                     code | (0 if explicit_light_sampling else 0x100) | (0x200 if render_mode == "rgb" else 0),
                                     table_path.encode() if table_path else None, jh_res, C.byref(self._h))
        if rc != 0:
            raise SsxError(rc, lib.ssh_last_error().decode())
        self.name = name

    @property
    def desc(self):
        return self._lib.ssh_scene_desc(self._h)

    def jh_model(self):
        """(res, scale[res], data[3*res^3*3]) of the Jakob-Hanika model in use, or None."""
        d = self.desc.contents
        if d.uplift != _capi.SSX_UPLIFT_JH:
            return None
        res = int(d.jh_res)
        return (res, np.ctypeslib.as_array(d.jh_scale, shape=(res,)).copy(),
                np.ctypeslib.as_array(d.jh_data, shape=(3 * res ** 3 * 3,)).copy())

    def xyza_to_srgba(self, xyza):
        xyza = np.ascontiguousarray(xyza, dtype=np.float32)
        out = np.empty_like(xyza)
        rc = self._lib.ssh_xyza_to_srgba(self._h, xyza.ctypes.data, out.ctypes.data, xyza.size // 4)
        if rc != 0:
            raise SsxError(rc, self._lib.ssh_last_error().decode())
        return out

    def color_values(self, name):
        buf = (C.c_float * 16)()
        n = self._lib.ssh_color_values(self._h, name.encode(), buf, 16)
        if n < 0:
            raise SsxError(n, "unknown colour table %r" % name)
        return np.array(buf[:n], dtype=np.float32)

    def close(self):
        if self._h:
            self._lib.ssh_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Renderer:
    """Renderer (src/renderer.hpp:14-82): render_start / render_stop / render_wait / is_rendering,
    public `framebuffer` and `scene`."""

    def __init__(self, options: Options):
        self.options = options
        self.scene = Scene(options.scene_name, options.observer, options.texture, options.light_scale, options.data_dir,
                           options.uplift, options.jh_res, options.jh_coeff_path, options.explicit_light_sampling, options.meng_grid_path, options.render_mode)
        self._lib = _capi.hip_lib()
        self._ctx = C.c_void_p()
        rc = self._lib.ssx_create(options.device, C.byref(self._ctx))
        if rc != 0:
            raise SsxError(rc, self._lib.ssx_last_error(None).decode())
        if options.jit_pass1 is not None:
            self._check(self._lib.ssx_set_jit(self._ctx, _capi.SSX_JIT_AT_UPLOAD if options.jit_pass1 else _capi.SSX_JIT_OFF))
        self._check(self._lib.ssx_upload_scene(self._ctx, self.scene.desc))
        W, H = options.res
        self.xyza = np.zeros((H, W, 4), dtype=np.float32)
        self.framebuffer = np.zeros((H, W, 4), dtype=np.float32)

    def _check(self, rc):
        if rc != 0:
            raise SsxError(rc, self._lib.ssx_last_error(self._ctx).decode())

    def params(self, **over):
        o = self.options
        p = _capi.SsxRenderParams()
        p.struct_size = C.sizeof(_capi.SsxRenderParams)
        p.width, p.height = o.res
        p.spp = o.spp
        p.indirect_only = int(o.indirect_only)
        p.no_explicit_light_sampling = int(not o.explicit_light_sampling)
        p.no_flat_field_correction = int(not o.flat_field_correction)
        p.tile_first, p.tile_stride = o.tile_first, o.tile_stride
        p.tile_skew = o.tile_skew
        p.spp_per_launch = o.spp_per_launch
        p.tile_major = int(o.tile_major)
        p.seed = o.seed
        for k, v in over.items():
            setattr(p, k, v)
        return p

    def render_start(self):
        self._check(self._lib.ssx_render_start(self._ctx, C.byref(self.params())))

    def render_stop(self):
        self._check(self._lib.ssx_render_stop(self._ctx))

    def is_rendering(self):
        return bool(self._lib.ssx_is_rendering(self._ctx))

    def progress(self):
        return float(self._lib.ssx_progress(self._ctx))

    def done_spp(self):
        """Samples per pixel accumulated so far (after render_stop: what the partial image is the mean of)."""
        return int(self._lib.ssx_done_spp(self._ctx))

    def done_tiles(self):
        """ssx_done_tiles: this device's tiles (ascending tile order) finished so far."""
        return int(self._lib.ssx_done_tiles(self._ctx))

    def render_wait(self):
        self._check(self._lib.ssx_render_wait(self._ctx, self.xyza.ctypes.data))
        self.framebuffer = self.scene.xyza_to_srgba(self.xyza)  # src/renderer.cpp:298
        if self.options.output_path:
            self.save(self.options.output_path)
        return self.framebuffer

    def render_device(self, d_ptr, stream=0, **over):
        """Enqueue the render on `stream` into the device buffer at d_ptr (W*H float4)."""
        p = self.params(**over)
        self._check(self._lib.ssx_render_device(self._ctx, C.byref(p), C.c_void_p(d_ptr), C.c_void_p(stream)))

    def render_device_wait(self):
        """ssx_render_device_wait: host wait for what render_device has queued for this context."""
        self._check(self._lib.ssx_render_device_wait(self._ctx))

    def set_jit(self, mode=_capi.SSX_JIT_AT_UPLOAD):
        """ssx_set_jit: SSX_JIT_OFF / SSX_JIT_AT_UPLOAD / SSX_JIT_BACKGROUND (True / False: at upload / off)."""
        self._check(self._lib.ssx_set_jit(self._ctx, int(mode)))

    def jit_status(self, wait_ms=0):
        """ssx_jit_status -> (state, message): SSX_JIT_STATE_NONE / _GENERIC_MEANWHILE / _SPECIALISED / _FAILED.  wait_ms != 0 asks
        for the compilation now and waits for it (< 0: until done)."""
        buf = C.create_string_buffer(1024)
        st = self._lib.ssx_jit_status(self._ctx, int(wait_ms), buf, len(buf))
        if st < -1:
            self._check(st)
        return st, buf.value.decode(errors="replace")

    def upload_scene_desc(self, desc):
        """Replace the scene by an arbitrary ssx_scene_desc (the flat description the C ABI takes)."""
        self._check(self._lib.ssx_upload_scene(self._ctx, C.byref(desc) if not hasattr(desc, "contents") else desc))

    def debug_eval(self, op, inputs, out_words):
        """ssx_debug_eval: `inputs` is an [n, in_words] array of 32-bit words (float32 or uint32 views);
        returns a uint32 array [n, out_words] (view it as float32 where the op returns floats)."""
        x = np.ascontiguousarray(inputs)
        assert x.ndim == 2 and x.dtype.itemsize == 4
        out = np.zeros((x.shape[0], out_words), dtype=np.uint32)
        self._check(self._lib.ssx_debug_eval(self._ctx, op, x.ctypes.data, x.shape[1], out.ctypes.data, out_words, x.shape[0]))
        return out

    def debug_sweep(self, op, lo=0, count=1 << 32):
        """ssx_debug_sweep -> (mismatches, op-specific maximum, [mismatching input bit patterns])"""
        res = (C.c_uint64 * 11)()
        self._check(self._lib.ssx_debug_sweep(self._ctx, op, lo, count, res))
        return int(res[0]), int(res[1]), [int(res[3 + k]) for k in range(min(int(res[2]), 8))]

    def debug_samples(self, **over):
        """ssx_debug_samples: per-sample (xyza [H, W, spp, 4], final PCG32 state [H, W, spp] uint64, levels [H, W, spp])."""
        p = self.params(**over)
        xyza = np.zeros((p.height, p.width, p.spp, 4), dtype=np.float32)
        state = np.zeros((p.height, p.width, p.spp), dtype=np.uint64)
        levels = np.zeros((p.height, p.width, p.spp), dtype=np.uint32)
        self._check(self._lib.ssx_debug_samples(self._ctx, C.byref(p), xyza.ctypes.data, state.ctypes.data, levels.ctypes.data))
        return xyza, state, levels

    def set_timing(self, enable=True):
        self._check(self._lib.ssx_set_timing(self._ctx, int(enable)))

    def get_timing(self):
        """Summed ms {generate, path, resolve, accumulate} of the launches since the last call."""
        ms = (C.c_float * 4)()
        self._check(self._lib.ssx_get_timing(self._ctx, ms))
        return dict(zip(("generate", "path", "resolve", "accumulate"), [float(x) for x in ms]))

    def kernel_info(self):
        v = [C.c_int() for _ in range(5)]
        self._check(self._lib.ssx_kernel_info(self._ctx, *[C.byref(x) for x in v]))
        return dict(zip(("vgprs", "sgprs", "lds_bytes", "scratch_bytes", "max_blocks_per_cu"), [x.value for x in v]))

    def plan_info(self):
        """What the calibration render at scene upload found: frames per sample, and where the fold runs."""
        f, k = C.c_float(), C.c_int()
        self._check(self._lib.ssx_plan_info(self._ctx, C.byref(f), C.byref(k)))
        variant = {0: "generic", 1: "cornell topology", 2: "plane topology", 3: "scene topology (compiled at upload)"}.get(self._lib.ssx_kernel_variant(self._ctx), "?")
        name = self._lib.ssx_kernel_name(self._ctx)
        left, pre = C.c_float(), C.c_int()
        if hasattr(self._lib, "ssx_calibration_info"):  # (an older build loaded through SSX_HIP_LIB_OVERRIDE for an A/B run has neither)
            self._check(self._lib.ssx_calibration_info(self._ctx, None, C.byref(left), C.byref(pre)))
        # where a sample's stream / camera ray / lambda_0 are made: the generate kernel, or -- kernels of the plane topology whose camera rays are
        # traced in the path loop (csrc/ssx_api.hip enqueue_front: fuse_gen) -- the path kernel's refill
        fused = (not pre.value) and variant == "plane topology" and not (os.environ.get("SSX_DEBUG_ENV") == "1" and os.environ.get("SSX_FUSE_GEN", "")[:1] == "0")
        return {"frames_per_sample": round(f.value, 3), "fold": "path kernel", "pass1": variant,
                "rays_left_per_sample": round(left.value, 3), "camera_rays": "pre-traced (generate kernel)" if pre.value else "path loop",
                "samples_made_in": "path kernel (refill)" if fused else "generate kernel",
                "kernel": name.decode() if name else None}

    def scratch_info(self):
        """Device scratch held: bytes of per-sample arrays (largest launch so far) and of the persistent waves' level logs."""
        a, b = C.c_uint64(), C.c_uint64()
        if hasattr(self._lib, "ssx_scratch_info"):
            self._check(self._lib.ssx_scratch_info(self._ctx, C.byref(a), C.byref(b)))
        return {"sample_bytes": a.value, "log_bytes": b.value}

    def sums_info(self):
        """ssx_sums_info: work units that parked their samples instead of waiting for their tile's turn, and how many of those the
        wave in front of them added (cumulative since the context was created)."""
        a, b = C.c_uint64(), C.c_uint64()
        if hasattr(self._lib, "ssx_sums_info"):
            self._check(self._lib.ssx_sums_info(self._ctx, C.byref(a), C.byref(b)))
        u = C.c_uint64()
        if hasattr(self._lib, "ssx_units_info"):
            self._check(self._lib.ssx_units_info(self._ctx, C.byref(u)))
        return {"units_parked": a.value, "units_chained": b.value, "units": u.value}

    def save(self, path):
        fb = np.ascontiguousarray(self.framebuffer, dtype=np.float32)
        rc = _capi.host_lib().ssh_save_image(path.encode(), fb.ctypes.data, fb.shape[1], fb.shape[0])
        if rc != 0:
            raise SsxError(rc, _capi.host_lib().ssh_last_error().decode())

    def close(self):
        if self._ctx:
            self._lib.ssx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
