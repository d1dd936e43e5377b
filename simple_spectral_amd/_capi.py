"""ctypes view of the C ABI (include/ssx.h) and of the host library (include/ssx_host.h)."""
import ctypes as C
import os

from . import build as _build


class SsxSpectrum(C.Structure):
    _fields_ = [("offset", C.c_uint32), ("n", C.c_uint32), ("low", C.c_float), ("high", C.c_float),
                ("delta_recip", C.c_float)]


class SsxVertex(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("st", C.c_float * 2)]


class SsxQuad(C.Structure):
    _fields_ = [("v00", SsxVertex), ("v10", SsxVertex), ("v11", SsxVertex), ("v01", SsxVertex),
                ("normal0", C.c_float * 3), ("normal1", C.c_float * 3),
                ("material", C.c_uint32), ("flags", C.c_uint32)]   # SSX_PRIM_LIGHT | SSX_PRIM_TRI


class SsxMaterial(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("albedo_mode", C.c_uint32), ("albedo_spectrum", C.c_uint32),
                ("albedo_texture", C.c_uint32), ("emission_spectrum", C.c_uint32)]


class SsxTexture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgb", C.POINTER(C.c_uint8))]


class SsxMengGrid(C.Structure):
    _fields_ = [("grid_w", C.c_uint32), ("grid_h", C.c_uint32), ("n_points", C.c_uint32), ("n_samples", C.c_uint32),
                ("sample_min", C.c_float), ("sample_max", C.c_float), ("xy_to_uv", C.c_float * 6),
                ("cells", C.POINTER(C.c_int32)), ("points", C.POINTER(C.c_float))]


class SsxSceneDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("reserved", C.c_uint32),
        ("pv_inv", C.c_double * 16), ("cam_pos", C.c_float * 3),
        ("lambda_min", C.c_float), ("lambda_step", C.c_float),
        ("spec_xbar", C.c_uint32), ("spec_ybar", C.c_uint32), ("spec_zbar", C.c_uint32),
        ("spec_basis_r", C.c_uint32), ("spec_basis_g", C.c_uint32), ("spec_basis_b", C.c_uint32),
        ("spectra", C.POINTER(SsxSpectrum)), ("n_spectra", C.c_uint32),
        ("samples", C.POINTER(C.c_float)), ("n_samples", C.c_uint32),
        ("materials", C.POINTER(SsxMaterial)), ("n_materials", C.c_uint32),
        ("quads", C.POINTER(SsxQuad)), ("n_quads", C.c_uint32),
        ("lights", C.POINTER(C.c_uint32)), ("n_lights", C.c_uint32),
        ("textures", C.POINTER(SsxTexture)), ("n_textures", C.c_uint32),
        ("srgb_to_linear", C.c_float * 256),
        ("uplift", C.c_uint32), ("jh_res", C.c_uint32),
        ("jh_scale", C.POINTER(C.c_float)), ("jh_data", C.POINTER(C.c_float)),
        ("meng", C.POINTER(SsxMengGrid)),
        ("cam_dir", C.c_float * 3), ("reserved2", C.c_uint32),
    ]


class SsxRenderParams(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32),
                ("indirect_only", C.c_uint32), ("tile_first", C.c_uint32), ("tile_stride", C.c_uint32),
                ("spp_per_launch", C.c_uint32), ("no_explicit_light_sampling", C.c_uint32), ("no_flat_field_correction", C.c_uint32),
                ("tile_major", C.c_uint32), ("tile_skew", C.c_uint32), ("seed", C.c_uint64)]


SSX_MODE_RGB, SSX_UPLIFT_OURS, SSX_UPLIFT_MENG, SSX_UPLIFT_JH = 0, 1, 2, 3
SSX_PRIM_LIGHT, SSX_PRIM_TRI = 1, 0x100
SSX_OK, SSX_ERR_DATA, SSX_ERR_ARG, SSX_ERR_SCENE, SSX_ERR_DEVICE, SSX_ERR_STATE = 0, -1, -2, -3, -10, -11
SSX_ABI_VERSION = 2   # include/ssx.h; checked against the library at load
SSX_JIT_OFF, SSX_JIT_AT_UPLOAD, SSX_JIT_BACKGROUND = 0, 1, 2
SSX_JIT_STATE_NONE, SSX_JIT_STATE_GENERIC_MEANWHILE, SSX_JIT_STATE_SPECIALISED, SSX_JIT_STATE_FAILED = 0, 1, 2, -1

# every symbol include/ssx.h and include/ssx_host.h declare (tests check the libraries export them)
HIP_SYMBOLS = ["ssx_create", "ssx_destroy", "ssx_upload_scene", "ssx_render_start", "ssx_render_stop",
               "ssx_is_rendering", "ssx_progress", "ssx_render_wait", "ssx_render_device", "ssx_last_error",
               "ssx_abi_version", "ssx_kernel_info", "ssx_plan_info", "ssx_set_timing", "ssx_get_timing",
               "ssx_device_framebuffer", "ssx_device_index", "ssx_read_framebuffer", "ssx_accumulate_peer",
               "ssx_debug_eval", "ssx_debug_samples", "ssx_debug_sweep", "ssx_kernel_variant", "ssx_kernel_name", "ssx_scratch_info", "ssx_calibration_info", "ssx_set_jit", "ssx_debug_pass1_source", "ssx_done_spp", "ssx_reduce_rccl", "ssx_sums_info", "ssx_rccl_groups_made", "ssx_jit_status", "ssx_jit_counters", "ssx_done_tiles", "ssx_render_device_wait", "ssx_units_info", "ssx_rccl_probe"]
(SSX_SWEEP_RCP, SSX_SWEEP_SQRT, SSX_SWEEP_INVERSESQRT, SSX_SWEEP_SIN, SSX_SWEEP_COS, SSX_SWEEP_ACOS, SSX_SWEEP_DIV_PI,
 SSX_SWEEP_RCP64, SSX_SWEEP_DIV_PAIRS, SSX_SWEEP_ACOS_SIN, SSX_SWEEP_SIN_PROOF, SSX_SWEEP_COS_PROOF, SSX_SWEEP_ACOS_PROOF) = range(1, 14)
# ssx_debug_eval ops (include/ssx.h)
(SSX_DBG_FMATH, SSX_DBG_SPHTRI, SSX_DBG_ARVO, SSX_DBG_SAMPLE_LIGHT, SSX_DBG_COSHEMI, SSX_DBG_TRACE, SSX_DBG_RAND_CHOICE,
 SSX_DBG_ALBEDO, SSX_DBG_FLUX_TO_XYZ, SSX_DBG_RAND_1F) = range(1, 11)
HOST_SYMBOLS = ["ssh_scene_create", "ssh_scene_create_ex", "ssh_scene_destroy", "ssh_scene_desc", "ssh_xyza_to_srgba", "ssh_save_image",
                "ssh_load_png_rgb8", "ssh_free", "ssh_color_values", "ssh_last_error"]

_hip = None
_host = None


def host_lib():
    global _host
    if _host is None:
        path = _build.HOST_LIB
        if not os.path.exists(path):
            _build.build_host()
        lib = C.CDLL(path)
        vp = C.c_void_p
        lib.ssh_last_error.restype = C.c_char_p
        lib.ssh_scene_create.argtypes = [C.c_char_p, C.c_char_p, C.c_int, vp, C.c_uint32, C.c_uint32, C.c_char_p,
                                         C.c_float, C.POINTER(vp)]
        lib.ssh_scene_create_ex.argtypes = [C.c_char_p, C.c_char_p, C.c_int, vp, C.c_uint32, C.c_uint32, C.c_char_p,
                                            C.c_float, C.c_uint32, C.c_char_p, C.c_uint32, C.POINTER(vp)]
        lib.ssh_scene_destroy.argtypes = [vp]
        lib.ssh_scene_desc.restype = C.POINTER(SsxSceneDesc)
        lib.ssh_scene_desc.argtypes = [vp]
        lib.ssh_xyza_to_srgba.argtypes = [vp, vp, vp, C.c_size_t]
        lib.ssh_save_image.argtypes = [C.c_char_p, vp, C.c_uint32, C.c_uint32]
        lib.ssh_load_png_rgb8.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint32),
                                          C.POINTER(C.c_uint32)]
        lib.ssh_free.argtypes = [vp]
        lib.ssh_color_values.argtypes = [vp, C.c_char_p, C.POINTER(C.c_float), C.c_int]
        _host = lib
    return _host


def mapped_hip_runtimes():
    """The distinct libamdhip64 files mapped into this process."""
    out = set()
    try:
        for ln in open("/proc/self/maps"):
            f = ln.split(None, 5)
            if len(f) == 6 and "libamdhip64.so" in os.path.basename(f[5].strip()):
                out.add(os.path.realpath(f[5].strip()))
    except OSError:
        pass
    return sorted(out)


def _elf_dynamic(path):
    """(SONAME, [NEEDED ...]) of a 64-bit little-endian ELF file, from its dynamic section; (None, []) if it cannot be read."""
    import struct
    try:
        with open(path, "rb") as f:
            eh = f.read(64)
            if eh[:6] != b"\x7fELF\x02\x01":
                return None, []
            shoff, = struct.unpack_from("<Q", eh, 0x28)
            shentsize, shnum = struct.unpack_from("<HH", eh, 0x3A)
            f.seek(shoff)
            sh = [struct.unpack_from("<IIQQQQIIQQ", f.read(shentsize)) for _ in range(shnum)]
            dyn = next((x for x in sh if x[1] == 6), None)           # SHT_DYNAMIC
            if dyn is None:
                return None, []
            strtab = sh[dyn[6]]                                      # sh_link: its string table
            f.seek(strtab[4]); names = f.read(strtab[5])
            f.seek(dyn[4]); raw = f.read(dyn[5])
        soname, needed = None, []
        for k in range(0, len(raw) - 15, 16):
            tag, val = struct.unpack_from("<qQ", raw, k)
            if tag in (1, 14):                                       # DT_NEEDED, DT_SONAME
                name = names[val:names.index(b"\0", val)].decode()
                if tag == 14:
                    soname = name
                else:
                    needed.append(name)
        return soname, needed
    except Exception:
        return None, []


def _one_hip_runtime_before_load(path):
    """ONE HIP runtime per process, whatever the import order (this used to be a usage rule).
    libssx_hip.so needs a libamdhip64 by soname ("libamdhip64.so.7" with ROCm 7); torch bundles its own copy.  Streams, events and
    device pointers are handed between torch and this library (bench.py, dist.py), which only works inside one runtime.  So:
      * a runtime is already mapped (torch was imported first, or the host linked one): the loader binds libssx_hip.so to it
        by soname -- nothing to do;
      * none is mapped, torch is installed AND its bundled copy carries the soname libssx_hip.so asks for: that copy is mapped NOW
        (RTLD_GLOBAL), so that libssx_hip.so binds to it and a later `import torch` finds its own runtime already in place;
      * no torch, or a torch built against another ROCm generation (another soname: preloading it would not satisfy the library's
        NEEDED entry and only put a second runtime into a renderer-only process): libssx_hip.so brings /opt/rocm's through its RUNPATH.
    _one_hip_runtime_after_load checks the outcome instead of trusting it."""
    if mapped_hip_runtimes():
        return
    try:
        import importlib.util
        wanted = [n for n in _elf_dynamic(path)[1] if n.startswith("libamdhip64.so")]
        spec = importlib.util.find_spec("torch")
        if wanted and spec and spec.submodule_search_locations:
            cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
            if os.path.exists(cand) and _elf_dynamic(cand)[0] == wanted[0]:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass  # (then the library's own RUNPATH decides; the check below still holds)


def _one_hip_runtime_after_load(path):
    rts = mapped_hip_runtimes()                                      # (/proc/self/maps: read here, once per process -- hip_lib() caches the library)
    if len(rts) > 1:
        raise RuntimeError("two HIP runtimes are mapped into this process (%s) after loading %s: streams and device pointers cannot be "
                           "handed between them.  Import simple_spectral_amd (or torch) before whatever loaded the other one, or "
                           "point LD_LIBRARY_PATH at one of them." % (", ".join(rts), path))


def hip_lib():
    """The HIP library.  There is no fallback: a missing library is an error."""
    global _hip
    if _hip is None:
        # SSX_HIP_LIB_OVERRIDE: A/B and profiling builds (tools/ab_bench.sh, tools/lanestat.py); like the library's own A/B
        # switches it is honoured only under the master switch SSX_DEBUG_ENV=1
        path = (os.environ.get("SSX_HIP_LIB_OVERRIDE") if os.environ.get("SSX_DEBUG_ENV") == "1" else None) or _build.HIP_LIB
        if not os.path.exists(path):
            raise RuntimeError("%s is missing: run `python -m simple_spectral_amd.build` (needs hipcc). "
                               "simple_spectral_amd has no CPU or PyTorch fallback path." % path)
        _one_hip_runtime_before_load(path)
        lib = C.CDLL(path)
        _one_hip_runtime_after_load(path)
        override = path != _build.HIP_LIB
        if not override and lib.ssx_abi_version() != SSX_ABI_VERSION:  # (an A/B library of an older revision is loaded as it is)
            raise RuntimeError("%s implements ABI version %d, this binding version %d (include/ssx.h): rebuild with `python -m simple_spectral_amd.build`"
                               % (path, lib.ssx_abi_version(), SSX_ABI_VERSION))
        vp = C.c_void_p
        lib.ssx_last_error.restype = C.c_char_p
        lib.ssx_last_error.argtypes = [vp]
        lib.ssx_create.argtypes = [C.c_int, C.POINTER(vp)]
        lib.ssx_destroy.argtypes = [vp]
        lib.ssx_destroy.restype = None
        lib.ssx_upload_scene.argtypes = [vp, C.POINTER(SsxSceneDesc)]
        lib.ssx_render_start.argtypes = [vp, C.POINTER(SsxRenderParams)]
        lib.ssx_render_stop.argtypes = [vp]
        lib.ssx_is_rendering.argtypes = [vp]
        lib.ssx_progress.argtypes = [vp]
        lib.ssx_progress.restype = C.c_float
        lib.ssx_render_wait.argtypes = [vp, vp]
        lib.ssx_render_device.argtypes = [vp, C.POINTER(SsxRenderParams), vp, vp]
        lib.ssx_kernel_info.argtypes = [vp] + [C.POINTER(C.c_int)] * 5
        lib.ssx_plan_info.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_int)]
        lib.ssx_set_timing.argtypes = [vp, C.c_int]
        lib.ssx_get_timing.argtypes = [vp, C.POINTER(C.c_float)]
        lib.ssx_device_framebuffer.argtypes = [vp]
        lib.ssx_device_framebuffer.restype = vp
        lib.ssx_device_index.argtypes = [vp]
        lib.ssx_read_framebuffer.argtypes = [vp, vp]
        lib.ssx_accumulate_peer.argtypes = [vp, vp, C.c_int, vp, C.c_uint32, C.c_uint32, vp]
        lib.ssx_debug_eval.argtypes = [vp, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32, C.c_uint32]
        lib.ssx_debug_samples.argtypes = [vp, C.POINTER(SsxRenderParams), vp, vp, vp]
        lib.ssx_debug_sweep.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64)]
        if not override or hasattr(lib, "ssx_scratch_info"):
            lib.ssx_calibration_info.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]
            lib.ssx_scratch_info.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        if not override or hasattr(lib, "ssx_set_jit"):
            lib.ssx_set_jit.argtypes = [vp, C.c_int]
            lib.ssx_done_spp.argtypes = [vp]
            lib.ssx_done_spp.restype = C.c_uint32
            lib.ssx_reduce_rccl.argtypes = [C.POINTER(vp), C.c_int, C.c_uint32, C.c_uint32]
            lib.ssx_debug_pass1_source.argtypes = [C.POINTER(C.c_uint8), C.c_uint32, C.c_char_p, C.c_char_p, C.c_size_t]
        if not override or hasattr(lib, "ssx_sums_info"):
            lib.ssx_sums_info.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        if hasattr(lib, "ssx_rccl_probe"):
            lib.ssx_rccl_probe.argtypes = [C.c_char_p, C.c_size_t]
        if hasattr(lib, "ssx_units_info"):  # (new in round 6: an older build loaded through SSX_HIP_LIB_OVERRIDE has none)
            lib.ssx_units_info.argtypes = [vp, C.POINTER(C.c_uint64)]
        if not override or hasattr(lib, "ssx_jit_status"):
            lib.ssx_jit_status.argtypes = [vp, C.c_int, C.c_char_p, C.c_size_t]
            lib.ssx_jit_counters.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
            lib.ssx_jit_counters.restype = None
            lib.ssx_rccl_groups_made.restype = C.c_uint64
            lib.ssx_render_device_wait.argtypes = [vp]
            lib.ssx_done_tiles.argtypes = [vp]
            lib.ssx_done_tiles.restype = C.c_uint32
        lib.ssx_kernel_variant.argtypes = [vp]
        lib.ssx_kernel_name.argtypes = [vp]
        lib.ssx_kernel_name.restype = C.c_char_p
        _hip = lib
    return _hip
