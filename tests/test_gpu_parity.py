"""Parity tests proper: the HIP path, called through the C ABI, against the CPU oracle and the
committed golden vectors.  Bar: BIT-EXACT float4 XYZA per pixel (integer-grade equality; the
north-star tolerance is 1e-4 relative, the build's contract is 0).  Run with -m gpu on MI355X."""
import ctypes as C
import os
import time

import numpy as np
import pytest

import oracle_lib as ol
from simple_spectral_amd import Options, Renderer, SsxError, _capi
from simple_spectral_amd import dist as sdist

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def gpu_render(**kw):
    r = Renderer(Options(**kw))
    r.render_start()
    r.render_wait()
    out = r.xyza.copy()
    return out, r


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_hip_library_is_the_one_running():
    lib = _capi.hip_lib()
    assert lib.ssx_abi_version() == _capi.SSX_ABI_VERSION == 2
    _, r = gpu_render(scene_name="cornell", res=(16, 16), spp=1)
    info = r.kernel_info()
    assert info["vgprs"] > 0 and info["lds_bytes"] > 4096  # the scene blob is staged in LDS


@pytest.mark.parametrize("scene,observer,W,H,spp,seed,io", [
    ("cornell", 1931, 64, 64, 4, 0, False),
    ("cornell-srgb", 1931, 64, 64, 4, 0, False),
    ("plane-srgb", 1931, 64, 64, 4, 0, False),
    ("cornell-srgb", 2006, 40, 40, 4, 9, False),
    ("cornell", 2006, 32, 32, 3, 1, False),
    ("cornell-srgb", 1931, 50, 37, 3, 2, False),      # ragged: edge tiles clipped (renderer.cpp:402)
    ("cornell-srgb", 1931, 7, 5, 9, 3, False),        # smaller than one tile
    ("cornell-srgb", 1931, 1, 1, 33, 4, False),
    ("cornell-srgb", 1931, 48, 32, 4, 5, True),       # --indirect-only
    ("plane-srgb", 1931, 33, 65, 2, 6, True),
])
def test_bit_exact_against_oracle(scene, observer, W, H, spp, seed, io):
    tex = None if scene == "cornell" else "test-img.png"
    got, _ = gpu_render(scene_name=scene, observer=observer, res=(W, H), spp=spp, seed=seed, indirect_only=io, texture=tex)
    ref = ol.Oracle(scene, observer=observer, texture=tex).render(W, H, spp, seed=seed, indirect_only=io)
    assert np.array_equal(bits(got), bits(ref))
    assert not np.isnan(got).any()


def test_config1_cornell_srgb_128_spp16_lizard_texture():
    """BASELINE.json configs[0] (cornell-srgb 128x128 spp=16 CIE1931) with the 512^2 lizard texture."""
    got, r = gpu_render(scene_name="cornell-srgb", res=(128, 128), spp=16, texture="crystal-lizard-512.png")
    o = ol.Oracle("cornell-srgb", texture="crystal-lizard-512.png")
    ref = o.render(128, 128, 16)
    assert np.array_equal(bits(got), bits(ref))
    # the XYZ -> sRGB store of renderer.cpp:298 (host side of the boundary)
    assert np.array_equal(bits(r.framebuffer), bits(o.to_srgba(ref)))
    # tolerance form of the north-star metric, for the record: max per-pixel relative dXYZ
    rel = np.abs(got[..., :3] - ref[..., :3]) / np.maximum(np.abs(ref[..., :3]), 1e-6)
    assert rel.max() <= 1e-4


def test_committed_goldens():
    g = np.load(os.path.join(HERE, "golden", "integrator_goldens.npz"))
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_goldens", os.path.join(HERE, "golden", "make_goldens.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    for name, scene, obs, tex, W, H, spp, seed, io in mg.CASES:
        got, _ = gpu_render(scene_name=scene, observer=obs, res=(W, H), spp=spp, seed=seed, indirect_only=io, texture=tex)
        assert np.array_equal(bits(got), bits(g[name + "__image"])), name


def test_launch_chunking_and_tile_partition_do_not_change_the_image():
    base, _ = gpu_render(scene_name="cornell-srgb", res=(72, 40), spp=12, seed=21, texture="test-img.png")
    for chunk in (1, 5, 12, 100):
        got, _ = gpu_render(scene_name="cornell-srgb", res=(72, 40), spp=12, seed=21, texture="test-img.png", spp_per_launch=chunk)
        assert np.array_equal(bits(got), bits(base)), chunk
    # tile_skew: the shared-out list with tile row ty rotated by ty * skew columns (diagonal instead of vertical stripes where the tile
    # row is a multiple of the device count: 72 / 8 = 9 tiles per row and world = 3; bench.py uses skew 1)
    for world, skew in ((2, 0), (3, 0), (8, 0), (3, 1), (8, 1), (4, 5), (3, 2 ** 31 + 5)):   # (the last: a skew whose 32-bit product with the tile row would wrap -- it is taken modulo the tile columns)
        parts = []
        for rank in range(world):
            got, _ = gpu_render(scene_name="cornell-srgb", res=(72, 40), spp=12, seed=21, texture="test-img.png",
                                tile_first=rank, tile_stride=world, tile_skew=skew)
            mask = sdist.tile_owner_mask(72, 40, rank, world, skew)
            assert mask.any() and not got[~mask].any()        # foreign tiles are exactly zero
            assert np.array_equal(bits(got[mask]), bits(base[mask]))
            parts.append(got)
        assert np.array_equal(bits(np.sum(parts, axis=0, dtype=np.float32)), bits(base))  # what the RCCL reduce computes
    m0, m1 = sdist.tile_owner_mask(72, 40, 0, 3, 0), sdist.tile_owner_mask(72, 40, 0, 3, 1)
    assert m0[:, :8].all() and not m1[:, :8].all()            # plain list: rank 0 owns the whole first tile column; rotated: a diagonal


def test_device_buffer_entry_point_matches_host_path():
    torch = pytest.importorskip("torch")
    r = Renderer(Options(scene_name="plane-srgb", res=(40, 24), spp=6, seed=2, texture="test-img.png"))
    ref = ol.Oracle("plane-srgb", texture="test-img.png").render(40, 24, 6, seed=2)
    out = torch.zeros((24, 40, 4), dtype=torch.float32, device="cuda")
    r.render_device(out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(bits(out.cpu().numpy()), bits(ref)), "ssx_render_device differs from the oracle"
    r.render_start(); r.render_wait()
    assert np.array_equal(bits(r.xyza), bits(ref)), "start/wait differs from the oracle"


def test_full_size_properties_config2():
    """BASELINE.json configs[1] size (cornell-srgb 512x512 spp=256): size-independent properties."""
    kw = dict(scene_name="cornell-srgb", res=(512, 512), spp=256, texture="crystal-lizard-512.png")
    a, r = gpu_render(**kw)
    b, _ = gpu_render(**kw)
    assert np.array_equal(bits(a), bits(b))                         # run-to-run determinism
    # linearity in the light: doubling the emission doubles every XYZ exactly (x2 commutes with rounding)
    c, _ = gpu_render(light_scale=60.0, **kw)
    assert np.array_equal(bits(c[..., :3]), bits(2.0 * a[..., :3])) and np.array_equal(bits(c[..., 3]), bits(a[..., 3]))
    # oracle spot checks at full spp on four 8x8 tiles (bit exact)
    o = ol.Oracle("cornell-srgb", texture="crystal-lizard-512.png")
    for (i0, j0) in ((0, 0), (256, 256), (504, 504), (120, 400)):
        ref = o.render(512, 512, 256, rect=(i0, j0, i0 + 8, j0 + 8))
        assert np.array_equal(bits(a[j0:j0 + 8, i0:i0 + 8]), bits(ref[j0:j0 + 8, i0:i0 + 8])), (i0, j0)
    # alpha = fraction of camera samples that hit anything (0.947)
    assert abs(float(a[..., 3].mean()) - 0.947) < 0.003
    # two-way tile split reassembles
    h0, _ = gpu_render(tile_first=0, tile_stride=2, **kw)
    h1, _ = gpu_render(tile_first=1, tile_stride=2, **kw)
    assert np.array_equal(bits(h0 + h1), bits(a))


def test_headline_config_whole_image_against_the_oracle():
    """BASELINE.json configs[1] -- the bench's workload: cornell-srgb 512x512 spp=256, CIE 1931, lizard texture, seed 0 -- the WHOLE image
    against the CPU oracle, all 1 048 576 floats bit for bit (67 M samples on the host's cores: ~20 s on 16).  bench.py checks tiles of the
    image it timed; this is the same image in full (reference: src/renderer.cpp:278-299, every pixel's ordered binary64 mean)."""
    got, _ = gpu_render(scene_name="cornell-srgb", res=(512, 512), spp=256, texture="crystal-lizard-512.png")
    ref = ol.Oracle("cornell-srgb", texture="crystal-lizard-512.png").render(512, 512, 256, nthreads=0)
    assert np.array_equal(bits(got), bits(ref)), "floats differing: %d" % int((bits(got) != bits(ref)).sum())
    rel = np.abs(got[..., :3] - ref[..., :3]) / np.maximum(np.abs(ref[..., :3]), 1e-6)   # the north-star form of the same statement
    assert rel.max() <= 1e-4


def test_async_interface_progress_and_stop():
    r = Renderer(Options(scene_name="cornell-srgb", res=(256, 256), spp=4096, spp_per_launch=8, texture="test-img.png"))
    r.render_start()
    assert r.is_rendering()
    with pytest.raises(SsxError) as e:
        r.render_start()                       # already rendering
    assert e.value.code == _capi.SSX_ERR_STATE
    t0 = time.time()
    while r.progress() == 0.0 and time.time() - t0 < 30:
        time.sleep(0.001)
    r.render_stop()                            # Renderer::render_stop: abort, image still produced
    r.render_wait()
    assert not r.is_rendering()
    assert 0.0 < r.progress() < 1.0
    assert np.isfinite(r.xyza).all()
    # the partial image is the mean over the samples DONE (not dimmed by done/total): alpha = hit fraction,
    # and it equals a complete render of that many samples bit for bit
    done = int(round(r.progress() * 4096))
    assert done % 8 == 0 and abs(float(r.xyza[..., 3].mean()) - 0.947) < 0.01
    full, _ = gpu_render(scene_name="cornell-srgb", res=(256, 256), spp=done, texture="test-img.png")
    assert np.array_equal(bits(r.xyza), bits(full))


def test_tile_major_render_keeps_finished_tiles_when_stopped():
    """the reference's stop semantics on offer.  ssx_render_params.tile_major walks through the tile list like
    the reference (src/renderer.cpp:340-409: tile (0,0) upwards, every tile to the full sample count).  A finished render is the same
    image bit for bit; a stopped one holds finished tiles at their FINAL value next to tiles that were never touched (returned as
    zeros; ssx_done_tiles tells the host where to leave its checkerboard, src/renderer.cpp:388-394), also on a device that owns
    every third tile."""
    import subprocess, sys
    # complete renders, several launches (3 tiles per launch, samples in ranges of 5): the oracle's bits
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np, oracle_lib as ol\nfrom simple_spectral_amd import Options, Renderer\n"
            "r = Renderer(Options(scene_name='cornell-srgb', res=(50, 37), spp=12, seed=4, texture='test-img.png', tile_major=True, spp_per_launch=5))\n"
            "r.render_start(); r.render_wait()\n"
            "ref = ol.Oracle('cornell-srgb', texture='test-img.png').render(50, 37, 12, seed=4)\n"
            "assert np.array_equal(r.xyza.view(np.uint32), ref.view(np.uint32)) and r.done_tiles() == 35 and r.done_spp() == 12\n" % (os.path.dirname(HERE), HERE))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SSX_DEBUG_ENV="1", SSX_TILES_PER_LAUNCH="3"), capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    # a stopped render
    for first, stride in ((0, 1), (2, 3)):
        o = dict(scene_name="cornell-srgb", res=(512, 512), spp=2048, seed=1, texture="test-img.png", tile_first=first, tile_stride=stride)
        r = Renderer(Options(tile_major=True, **o))
        owned = (4096 - first + stride - 1) // stride
        r.render_start()
        t0 = time.time()
        while r.done_tiles() == 0 and r.is_rendering() and time.time() - t0 < 60:
            time.sleep(0.002)
        r.render_stop()
        r.render_wait()
        done = r.done_tiles()
        assert 0 < done < owned, (done, owned)          # (512 x 512 x 2048 samples take ~170 ms; the first launch ~10 ms)
        full, _ = gpu_render(**o)                        # the same render, finished (sample-major)
        tile = (np.arange(512)[:, None] // 8) * 64 + np.arange(512)[None, :] // 8
        finished = (tile % stride == first) & (tile // stride < done)
        assert np.array_equal(bits(r.xyza[finished]), bits(full[finished])) and not r.xyza[~finished].any()
        assert float(r.xyza[finished][:, 3].mean()) > 0.5


def test_error_codes():
    lib = _capi.hip_lib()
    ctx = C.c_void_p()
    assert lib.ssx_create(99, C.byref(ctx)) == _capi.SSX_ERR_ARG
    assert lib.ssx_create(0, C.byref(ctx)) == 0
    p = _capi.SsxRenderParams(); p.struct_size = C.sizeof(p); p.width = p.height = 8; p.spp = 1; p.tile_stride = 1
    assert lib.ssx_render_start(ctx, C.byref(p)) == _capi.SSX_ERR_STATE      # no scene uploaded
    assert b"no scene" in lib.ssx_last_error(ctx)
    assert lib.ssx_render_wait(ctx, None) == _capi.SSX_ERR_STATE
    from simple_spectral_amd.renderer import Scene
    s = Scene("cornell")
    assert lib.ssx_upload_scene(ctx, s.desc) == 0
    p.spp = 0
    assert lib.ssx_render_start(ctx, C.byref(p)) == _capi.SSX_ERR_ARG
    p.spp = 1; p.tile_first = 3; p.tile_stride = 2
    assert lib.ssx_render_start(ctx, C.byref(p)) == _capi.SSX_ERR_ARG
    bad = _capi.SsxSceneDesc(); bad.struct_size = 12
    assert lib.ssx_upload_scene(ctx, C.byref(bad)) == _capi.SSX_ERR_ARG
    lib.ssx_destroy(ctx)


def same_or_both_nan(a, b):
    """Bit equality, except that NaN matches NaN of any payload (black texels make the reference's
    Jakob-Hanika path divide by zero, rgb2spec.c:88-91, so both sides hold NaN there)."""
    return ((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))).all()


@pytest.mark.parametrize("scene", ["cornell-srgb", "plane-srgb"])
def test_jakob_hanika_uplift_bit_exact(scene):
    """BASELINE configs[2] names the Jakob-Hanika uplift (RENDER_MODE_SPECTRAL_ALGNUM 3): texels go
    through rgb2spec_fetch / rgb2spec_eval_precise instead of the basis spectra."""
    r = Renderer(Options(scene_name=scene, res=(48, 40), spp=5, seed=4, texture="test-img.png", uplift="jh", jh_res=16))
    r.render_start(); r.render_wait()
    ref = ol.Oracle(scene, texture="test-img.png", jh=r.scene.jh_model()).render(48, 40, 5, seed=4)
    assert same_or_both_nan(r.xyza, ref)
    assert np.isfinite(ref).mean() > 0.9
    ours = ol.Oracle(scene, texture="test-img.png").render(48, 40, 5, seed=4)
    assert not np.array_equal(bits(ours), bits(ref))        # the variant really changes textured pixels


def test_config3_named_combination_cornell_with_jakob_hanika():
    """BASELINE configs[2] as named: scene "cornell" (spectral materials, no texture) built with the
    Jakob-Hanika uplift -- arithmetically a no-op there, but the JH blob layout (scale[] in LDS, table
    pointer) runs on that scene; small image against the oracle, then the full 512x512 spp=1024 size with
    oracle spot tiles at full spp."""
    r = Renderer(Options(scene_name="cornell", res=(48, 40), spp=5, seed=4, uplift="jh", jh_res=16))
    r.render_start(); r.render_wait()
    o = ol.Oracle("cornell", jh=r.scene.jh_model())
    assert np.array_equal(bits(r.xyza), bits(o.render(48, 40, 5, seed=4)))
    assert np.array_equal(bits(r.xyza), bits(ol.Oracle("cornell").render(48, 40, 5, seed=4)))   # no textured surface: same image
    a, _ = gpu_render(scene_name="cornell", res=(512, 512), spp=1024, uplift="jh", jh_res=16)
    for (i0, j0) in ((248, 248), (40, 464)):
        ref = o.render(512, 512, 1024, rect=(i0, j0, i0 + 8, j0 + 8))
        assert np.array_equal(bits(a[j0:j0 + 8, i0:i0 + 8]), bits(ref[j0:j0 + 8, i0:i0 + 8])), (i0, j0)
    assert np.isfinite(a).all()


def test_config4_per_gpu_share_plane_srgb_1024_spp4096_tile_split():
    """BASELINE configs[3]: plane-srgb 1024x1024 spp=4096 on 4 GPUs = every rank renders the tiles
    t % 4 == rank at spp 4096.  One rank's share at full size: foreign tiles exactly zero, two of its own
    tiles against the oracle at full spp."""
    W = H = 1024
    rank, world = 1, 4
    a, _ = gpu_render(scene_name="plane-srgb", res=(W, H), spp=4096, texture="crystal-lizard-512.png", tile_first=rank, tile_stride=world)
    mask = sdist.tile_owner_mask(W, H, rank, world)
    assert not a[~mask].any() and np.isfinite(a).all()
    assert abs(float(a[mask][:, 3].mean()) - 1.0) < 1e-6      # every camera ray hits the plane or the light box
    o = ol.Oracle("plane-srgb", texture="crystal-lizard-512.png")
    for tile in (1 + 4 * 1000, 1 + 4 * 3000):                  # tiles of this rank (tile % 4 == 1)
        tx, ty = tile % (W // 8), tile // (W // 8)
        i0, j0 = tx * 8, ty * 8
        ref = o.render(W, H, 4096, rect=(i0, j0, i0 + 8, j0 + 8))
        assert np.array_equal(bits(a[j0:j0 + 8, i0:i0 + 8]), bits(ref[j0:j0 + 8, i0:i0 + 8])), tile


def test_config5_per_gpu_share_cornell_srgb_2048_cie2006():
    """BASELINE configs[4]: cornell-srgb 2048x2048 spp=16384, CIE 2006 observer, on 8 GPUs = every rank
    renders the tiles t % 8 == rank at spp 16384 (8.6 G samples per GPU).  One rank's share at full size."""
    W = H = 2048
    rank, world = 5, 8
    a, _ = gpu_render(scene_name="cornell-srgb", observer=2006, res=(W, H), spp=16384, texture="crystal-lizard-512.png", tile_first=rank, tile_stride=world)
    mask = sdist.tile_owner_mask(W, H, rank, world)
    assert not a[~mask].any() and np.isfinite(a).all()
    assert abs(float(a[mask][:, 3].mean()) - 0.947) < 0.02   # this rank's tiles are the columns 5, 13, 21, ...: a biased sample of the image
    o = ol.Oracle("cornell-srgb", observer=2006, texture="crystal-lizard-512.png")
    tile = rank + 8 * 4100
    tx, ty = tile % (W // 8), tile // (W // 8)
    i0, j0 = tx * 8, ty * 8
    ref = o.render(W, H, 16384, rect=(i0, j0, i0 + 8, j0 + 8), nthreads=1)
    assert np.array_equal(bits(a[j0:j0 + 8, i0:i0 + 8]), bits(ref[j0:j0 + 8, i0:i0 + 8]))


@pytest.mark.parametrize("scene,W,H,spp", [("cornell-srgb", 128, 128, 16), ("plane-srgb", 192, 192, 4)])
def test_4096_texture_footprint(scene, W, H, spp):
    """The reference's -srgb scenes open a 4096x4096 texture (src/scene.cpp:292,357; 48 MiB of RGB8, beyond
    the 32 MiB of aggregate L2); its blob is missing from the repository, so a seeded procedural texture of
    that size stands in (SURVEY.md section 8(d) iii)."""
    from simple_spectral_amd import textures
    tex = textures.procedural_texture(4096, 1)
    got, r = gpu_render(scene_name=scene, res=(W, H), spp=spp, seed=2, texture="procedural:4096")   # the host library's generator
    ref = ol.Oracle(scene, texture=tex).render(W, H, spp, seed=2)                                    # the numpy generator
    assert np.array_equal(bits(got), bits(ref))
    small = ol.Oracle(scene, texture="crystal-lizard-512.png").render(W, H, spp, seed=2)
    assert not np.array_equal(bits(small), bits(ref))


def test_jakob_hanika_lizard_texture_config1_shape():
    r = Renderer(Options(scene_name="cornell-srgb", res=(128, 128), spp=16, texture="crystal-lizard-512.png", uplift="jh", jh_res=32))
    r.render_start(); r.render_wait()
    ref = ol.Oracle("cornell-srgb", texture="crystal-lizard-512.png", jh=r.scene.jh_model()).render(128, 128, 16)
    assert same_or_both_nan(r.xyza, ref)


def test_mirror_material_bit_exact():
    """MaterialMirror (src/material.cpp:146-167): delta BSDF, evaluate -> 0, sample -> reflect with
    pdf = +inf (then n_dot_l = pdf = 1, src/renderer.cpp:236-243).  No built-in scene selects it in
    the explicit-light-sampling build, so the textured quad of plane-srgb and the blocks of the
    Cornell box are switched to mirrors through the scene description on both sides."""
    import ctypes as C
    for scene, pick in (("plane-srgb", lambda q: q == 0), ("cornell-srgb", lambda q: q >= 9)):
        r = Renderer(Options(scene_name=scene, res=(40, 32), spp=6, seed=8, texture="test-img.png"))
        d = r.scene.desc.contents
        mats = (_capi.SsxMaterial * d.n_materials)(*[d.materials[i] for i in range(d.n_materials)])
        o = ol.Oracle(scene, texture="test-img.png")
        changed = set()
        for q in range(d.n_quads):
            if pick(q):
                m = d.quads[q].material
                mats[m].kind = 1  # SSX_MTL_MIRROR
                changed.add(m)
                assert o.lib.orc_scene_set_material_kind(o.scene, o.lib.orc_scene_quad_material(o.scene, q), 1) == 0
        d2 = _capi.SsxSceneDesc.from_buffer_copy(d)
        d2.materials = C.cast(mats, C.POINTER(_capi.SsxMaterial))
        r._check(r._lib.ssx_upload_scene(r._ctx, C.byref(d2)))
        r.render_start(); r.render_wait()
        ref = o.render(40, 32, 6, seed=8)
        assert np.array_equal(bits(r.xyza), bits(ref)), scene
        base = ol.Oracle(scene, texture="test-img.png").render(40, 32, 6, seed=8)
        assert not np.array_equal(bits(base), bits(ref)) and changed


@pytest.mark.parametrize("scene", ["plane-srgb", "cornell-srgb"])
def test_without_explicit_light_sampling_bit_exact(scene):
    """The integrator the reference compiles without EXPLICIT_LIGHT_SAMPLING (src/stdafx.hpp:44):
    emission at every hit, no next-event estimation, rays down to depth MAX_DEPTH-1; plane-srgb's
    textured quad is then a MaterialMirror (src/scene.cpp:346-355)."""
    r = Renderer(Options(scene_name=scene, res=(40, 32), spp=6, seed=3, texture="test-img.png", explicit_light_sampling=False))
    r.render_start(); r.render_wait()
    o = ol.Oracle(scene, texture="test-img.png")
    if scene == "plane-srgb":
        assert o.lib.orc_scene_set_material_kind(o.scene, o.lib.orc_scene_quad_material(o.scene, 0), 1) == 0
    ref = o.render(40, 32, 6, seed=3, els=False)
    assert np.array_equal(bits(r.xyza), bits(ref))
    assert not np.array_equal(bits(o.render(40, 32, 6, seed=3)), bits(ref))


@pytest.fixture(scope="module")
def meng_grid(tmp_path_factory):
    """The Meng et al. grid, read out of the reference's own header through oracle/_ref/libref_meng.so
    (built in place by `make -C oracle ref`; the .so travels to the GPU box, the header does not)."""
    import ref_lib
    from simple_spectral_amd import meng
    if ref_lib.meng() is None:
        pytest.skip("oracle/_ref/libref_meng.so not built (needs /root/reference at build time)")
    table = ref_lib.meng_table()
    path = str(tmp_path_factory.mktemp("meng") / "grid.bin")
    meng.save_table(path, table)
    return table, path


@pytest.mark.parametrize("scene,texture,W,H,spp", [("cornell-srgb", "test-img.png", 48, 40, 5), ("plane-srgb", "test-img.png", 48, 40, 5),
                                                   ("cornell-srgb", "crystal-lizard-512.png", 128, 128, 8)])
def test_meng_uplift_bit_exact(meng_grid, scene, texture, W, H, spp):
    """RENDER_MODE_SPECTRAL_ALGNUM 2 (src/util/color.cpp:175-201): texels go through Meng et al.'s
    spectrum_xyz_to_p.  The oracle's restatement of it is itself pinned bit for bit against the
    reference's function (tests/test_ref_pins.py), so this is HIP == reference code for the uplift."""
    table, path = meng_grid
    r = Renderer(Options(scene_name=scene, res=(W, H), spp=spp, seed=4, texture=texture, uplift="meng", meng_grid_path=path))
    r.render_start(); r.render_wait()
    orc = ol.Oracle(scene, texture=texture, meng=table)
    ref = orc.render(W, H, spp, seed=4)
    assert np.array_equal(bits(r.xyza), bits(ref))
    assert np.isfinite(ref).all()
    ours = ol.Oracle(scene, texture=texture).render(W, H, spp, seed=4)
    assert not np.array_equal(bits(ours), bits(ref))        # the variant really changes textured pixels
    # output transform of the variant (color.cpp:243-254), host vs oracle
    assert np.array_equal(bits(r.framebuffer), bits(orc.to_srgba(ref)))


def test_meng_grid_errors(meng_grid, tmp_path):
    table, path = meng_grid
    with pytest.raises(SsxError) as e:
        Renderer(Options(scene_name="cornell-srgb", texture="test-img.png", uplift="meng", meng_grid_path=str(tmp_path / "missing.bin")))
    assert e.value.code == _capi.SSX_ERR_DATA
    with pytest.raises(SsxError) as e:
        Renderer(Options(scene_name="cornell-srgb", texture="test-img.png", uplift="meng", meng_grid_path=path, observer=2006))
    assert e.value.code == _capi.SSX_ERR_SCENE                # stdafx.hpp:107-109


@pytest.mark.parametrize("scene,texture,W,H,spp,io,els", [
    ("cornell", None, 64, 48, 6, False, True), ("cornell-srgb", "test-img.png", 48, 40, 5, False, True),
    ("cornell-srgb", "crystal-lizard-512.png", 128, 128, 8, True, True), ("plane-srgb", "test-img.png", 48, 40, 5, False, True),
    ("plane-srgb", "test-img.png", 40, 40, 4, False, False), ("cornell-srgb", "test-img.png", 33, 27, 3, False, False)])
def test_rgb_render_mode_bit_exact(scene, texture, W, H, spp, io, els):
    """RENDER_MODE_RGB (src/stdafx.hpp:91-93): the integrator carries linear RGB -- no wavelength draw
    (src/renderer.cpp:134-143), texels as they are (src/material.cpp:61-63), RGB scene constants
    (src/scene.cpp:69-82,106,314,341), plain mean of the samples (src/renderer.cpp:300-304), sRGB
    transfer only on output (:306).  `xyza` then holds lRGB + alpha."""
    kw = dict(texture=texture) if texture else {}
    r = Renderer(Options(scene_name=scene, res=(W, H), spp=spp, seed=6, render_mode="rgb", indirect_only=io, explicit_light_sampling=els, **kw))
    r.render_start(); r.render_wait()
    orc = ol.Oracle(scene, rgb=True, **kw)
    if not els and scene == "plane-srgb":
        orc.lib.orc_scene_set_material_kind(orc.scene, orc.lib.orc_scene_quad_material(orc.scene, 0), 1)   # mirror (scene.cpp:346-355)
    ref = orc.render(W, H, spp, seed=6, indirect_only=io, els=els)
    assert np.array_equal(bits(r.xyza), bits(ref))
    assert np.isfinite(ref).all() and ref[..., :3].max() > 0
    assert np.array_equal(bits(r.framebuffer), bits(orc.to_srgba(ref)))
    if scene != "plane-srgb" or els:
        spectral = ol.Oracle(scene, **kw).render(W, H, spp, seed=6, indirect_only=io, els=els)
        assert not np.array_equal(bits(spectral), bits(ref))


@pytest.mark.parametrize("scene,io", [("cornell-srgb", False), ("plane-srgb", False), ("cornell", True)])
def test_without_flat_field_correction_bit_exact(scene, io):
    """The reference's other compile-time switch of the estimator, FLAT_FIELD_CORRECTION (src/stdafx.hpp:55): without it
    flux = radiance * dot(camera_ray_dir, camera.dir) (src/renderer.cpp:262-266).  A render parameter here."""
    W, H, spp = 40, 32, 8
    tex = None if scene == "cornell" else "test-img.png"
    got, _ = gpu_render(scene_name=scene, res=(W, H), spp=spp, seed=8, texture=tex, indirect_only=io, flat_field_correction=False)
    orc = ol.Oracle(scene, texture=tex)
    ref = orc.render(W, H, spp, seed=8, indirect_only=io, flat_field=False)
    assert np.array_equal(bits(got), bits(ref))
    assert not np.array_equal(bits(ref), bits(orc.render(W, H, spp, seed=8, indirect_only=io)))   # (and it is another image)


def test_calibration_and_device_scratch():
    """ssx_upload_scene renders 64x64x4 samples of the scene to count the continued levels per sample (unit size, byte
    accounting of the benchmark).  The levels of the recursion live in logs owned by the persistent waves and recycled
    while the kernel runs: the device scratch of a render is 48 bytes per sample in the launch plus
    logs whose size does not depend on the launch -- BASELINE configs[1] (512^2 x 256 spp, 67 M samples: 41.6 GB with the
    per-sample level arrays of round 2) fits in 4.5 GB."""
    import torch
    r = Renderer(Options(scene_name="plane-srgb", res=(16, 16), spp=1, texture="test-img.png"))
    info = r.plan_info()
    assert info["fold"] == "path kernel" and 0.8 < info["frames_per_sample"] <= 1.0      # S = 2 where the plane is hit: one continued level
    r = Renderer(Options(scene_name="cornell-srgb", res=(512, 512), spp=256, texture="crystal-lizard-512.png"))
    info = r.plan_info()
    assert info["fold"] == "path kernel" and 3.0 < info["frames_per_sample"] < 4.5      # interactions per sample - 1
    logs0 = r.scratch_info()["log_bytes"]
    waves = torch.cuda.get_device_properties(0).multi_processor_count * 16
    assert logs0 == waves * 2 * 2 * 128 * 582                                            # units of 4 samples per pixel: two cohorts
    out = torch.zeros((512, 512, 4), device="cuda")
    r.render_device(out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    sc = r.scratch_info()
    assert sc["log_bytes"] == logs0 and sc["sample_bytes"] == 512 * 512 * 256 * 48
    assert sc["log_bytes"] + sc["sample_bytes"] < 4.5e9
    assert float(out[..., 3].mean()) > 0.9


def test_camera_rays_pretraced_where_rays_leave_the_scene():
    """Where rays leave the scene often (Cornell box: 0.87 per sample through the open front) the generate kernel traces
    the camera rays, coherently, and the path loop starts every sample at its first hit, so that no lane idles through a
    shading; where they do not (plane-srgb: the light box is closed) the path loop traces camera rays like all others.
    The calibration render at scene upload decides; SSX_PRE_HITS=0/1 forces a mode.  Same bits either way: subprocesses
    repeat the oracle comparisons with each scene in the mode it does not take by itself."""
    import subprocess, sys
    info = Renderer(Options(scene_name="cornell-srgb", res=(8, 8), spp=1, texture="test-img.png")).plan_info()
    assert info["camera_rays"].startswith("pre-traced") and 0.6 < info["rays_left_per_sample"] < 1.1
    info = Renderer(Options(scene_name="plane-srgb", res=(8, 8), spp=1, texture="test-img.png")).plan_info()
    assert info["camera_rays"] == "path loop" and info["rays_left_per_sample"] < 0.05
    for mode in ("0", "1"):
        env = dict(os.environ, SSX_DEBUG_ENV="1", SSX_PRE_HITS=mode)
        out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k",
                              "bit_exact_against_oracle or config1 or launch_chunking or per_sample or without_explicit or mirror"], env=env, capture_output=True, text=True, cwd=os.path.dirname(HERE))
        assert out.returncode == 0, out.stdout[-3000:]


def test_pixel_sums_chain_through_the_units_of_a_tile():
    """The binary64 pixel sums (src/renderer.cpp:292-295: ascending k) are continued inside the path kernel: the units of
    a tile -- groups of consecutive samples, folded by whichever waves took them -- are added in k order, and no wave waits:
    a unit that finishes before its turn parks its samples, and the wave in front of it adds them (ssx_kernels.hip unit_fold).
    A tiny image with many samples per pixel makes the chains long and the waves many (hundreds of units of one tile in flight
    at once); also split over several launches (the sums continue) and on a ragged tile.  ssx_sums_info proves that the renders
    went through the parked and the chained branch."""
    for (W, H, spp, chunk) in ((16, 8, 2048, 0), (8, 8, 1536, 500), (11, 5, 777, 0)):
        got, r = gpu_render(scene_name="cornell-srgb", res=(W, H), spp=spp, seed=33, texture="test-img.png", spp_per_launch=chunk)
        ref = ol.Oracle("cornell-srgb", texture="test-img.png").render(W, H, spp, seed=33)
        assert np.array_equal(bits(got), bits(ref)), (W, H, spp, chunk)
        info = r.sums_info()
        assert info["units_parked"] > 10 and 0 < info["units_chained"] <= info["units_parked"], info
    # One rank's share of BASELINE configs[1] on 8 GPUs in miniature -- every eighth tile, eight times the samples per pixel:
    # 8 tiles x 128 units each, all 1024 in flight at once -- on the specialised kernel, three times over (the order in which the
    # units finish differs from run to run, the image must not), foreign tiles exactly zero.
    import torch
    r = Renderer(Options(scene_name="cornell-srgb", res=(64, 64), spp=512, seed=7, texture="test-img.png", tile_first=3, tile_stride=8))
    ref = ol.Oracle("cornell-srgb", texture="test-img.png").render(64, 64, 512, seed=7)
    mask = sdist.tile_owner_mask(64, 64, 3, 8)
    ref[~mask] = 0.0
    out = torch.zeros((64, 64, 4), device="cuda")
    for _ in range(3):
        r.render_device(out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(bits(out.cpu().numpy()), bits(ref))
    assert r.sums_info()["units_parked"] > 100
    # Units that run neck and neck: 128 x 128 at 16 spp is 4096 units of ONE sample per pixel, all handed out at once, the 16 of a
    # tile finishing within microseconds of each other -- every hand-over of the sums happens while the predecessor's stores are
    # still on their way (this case lost samples before the hand-over waited for them).  40 renders, one oracle.
    r = Renderer(Options(scene_name="cornell-srgb", res=(128, 128), spp=16, seed=2, texture="test-img.png"))
    ref = ol.Oracle("cornell-srgb", texture="test-img.png").render(128, 128, 16, seed=2)
    out = torch.zeros((128, 128, 4), device="cuda")
    for _ in range(40):
        r.render_device(out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(bits(out.cpu().numpy()), bits(ref))


def test_many_units_per_wave_parity_and_determinism():
    """Rare-event guard for the persistent-wave machinery (unit rotation, the shadow-ray queue's
    read-modify-write of frames and records, the fold's view of this wave's own stores): a render with
    ~10 units per wave against the oracle bit for bit, and a larger one three times over.
    tools/stress_parity.py does the same at 1.6 G samples."""
    import torch
    r = Renderer(Options(scene_name="cornell-srgb", res=(512, 512), spp=32, seed=11, texture="crystal-lizard-512.png"))
    r.render_start(); r.render_wait()
    ref = ol.Oracle("cornell-srgb", texture="crystal-lizard-512.png").render(512, 512, 32, seed=11)
    assert np.array_equal(bits(r.xyza), bits(ref))
    r = Renderer(Options(scene_name="cornell", res=(1024, 1024), spp=96, seed=5))
    outs = []
    for _ in range(3):
        out = torch.zeros((1024, 1024, 4), device="cuda")
        r.render_device(out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
    assert np.array_equal(bits(outs[0]), bits(outs[1])) and np.array_equal(bits(outs[0]), bits(outs[2]))
    assert np.isfinite(outs[0]).all()


def test_pass1_variants_specialised_for_builtin_topologies_generic_otherwise():
    """The built-in scenes run the kernels whose pass 1 is specialised to their mesh topology (every test above
    therefore covers those); a scene with another sharing pattern -- here: one corner of the Cornell box moved
    apart from its twin -- runs the generic loop.  With SSX_GENERIC_KERNEL set the built-in scenes run the generic
    loop too: a subprocess repeats the oracle comparisons that way."""
    import subprocess, sys
    import custom_scene as cs
    r = Renderer(Options(scene_name="cornell-srgb", res=(8, 8), spp=1, texture="test-img.png"))
    assert r.plan_info()["pass1"] == "cornell topology"
    assert Renderer(Options(scene_name="plane-srgb", res=(8, 8), spp=1, texture="test-img.png")).plan_info()["pass1"] == "plane topology"
    c = cs.CustomScene("cornell-srgb")
    pos, st, m = c.quads[0]
    pos = pos.copy(); pos[0, 0] += 1.0                     # the floor's first corner no longer coincides with the left wall's
    c.quads[0] = (pos, st, m)
    orc = c.oracle()
    r.upload_scene_desc(c.desc(orc))
    assert r.plan_info()["pass1"] == "generic"
    r.options.res = (40, 32); r.options.spp = 4; r.options.seed = 3
    r.xyza = np.zeros((32, 40, 4), dtype=np.float32)
    r.render_start(); r.render_wait()
    assert np.array_equal(bits(r.xyza), bits(orc.render(40, 32, 4, seed=3)))
    env = dict(os.environ, SSX_DEBUG_ENV="1", SSX_GENERIC_KERNEL="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k",
                          "bit_exact_against_oracle or config1 or many_units or jakob_hanika_uplift"], env=env, capture_output=True, text=True, cwd=os.path.dirname(HERE))
    assert out.returncode == 0, out.stdout[-3000:]


def test_pass1_compiled_at_upload_for_any_topology():
    """with ssx_set_jit the scene that matches no built-in topology -- the Cornell box with one corner
    moved apart from its twin -- gets ITS straight-line pass 1: generated at upload (csrc/ssx_jit.h, the generator of
    tools/gen_pass1.py in C++), compiled with hipRTC, launched from the module.  Same bits as the generic loop and the
    oracle; a second upload of the same pattern reuses the code; a scene with triangles or more than 32 primitives stays
    generic.  (tools/jit_rate.py measures the three rates side by side.)"""
    import time
    import crafted
    import custom_scene as cs
    c = cs.CustomScene("cornell-srgb")
    pos, st, m = c.quads[0]
    pos = pos.copy(); pos[0, 0] += 1.0
    c.quads[0] = (pos, st, m)
    orc = c.oracle()
    r = Renderer(Options(scene_name="cornell-srgb", res=(40, 32), spp=4, seed=3, texture="test-img.png", jit_pass1=True))
    assert r.plan_info()["pass1"] == "cornell topology"                       # a built-in pattern keeps its built-in kernel
    t = time.time(); r.upload_scene_desc(c.desc(orc)); first = time.time() - t
    info = r.plan_info()
    assert info["pass1"].startswith("scene topology") and info["kernel"] == "ssx_render_kernel_jit"
    assert r.kernel_info()["vgprs"] <= 128 and r.kernel_info()["scratch_bytes"] == 0
    r.render_start(); r.render_wait()
    ref = orc.render(40, 32, 4, seed=3)
    assert np.array_equal(bits(r.xyza), bits(ref))
    xyza, state, _ = r.debug_samples()
    ref_xyza, ref_state, _st = orc.samples(40, 32, 4, seed=3)
    assert np.array_equal(state, ref_state) and np.array_equal(bits(xyza), bits(ref_xyza))
    t = time.time(); r.upload_scene_desc(c.desc(orc)); again = time.time() - t   # same pattern: no second compilation
    assert again < 0.5 * first or again < 1.0, (first, again)
    # CIE 2006 tables + this pattern: the narrow-queue kernel of the same module
    c6 = cs.CustomScene("cornell-srgb", observer=2006)
    c6.quads[0] = (pos, st, m)
    orc6 = c6.oracle()
    r6 = Renderer(Options(scene_name="cornell-srgb", res=(24, 24), spp=3, seed=5, texture="test-img.png", observer=2006, jit_pass1=True))
    r6.upload_scene_desc(c6.desc(orc6))
    assert r6.plan_info()["kernel"] == "ssx_render_kernel_jit_nq"
    r6.render_start(); r6.render_wait()
    assert np.array_equal(bits(r6.xyza), bits(orc6.render(24, 24, 3, seed=5)))
    # not candidates: triangles, more than 32 primitives
    ct = crafted.triangle_scene()
    rt = Renderer(Options(scene_name="cornell", res=(8, 8), spp=1, jit_pass1=True))
    rt.upload_scene_desc(ct.desc(ct.oracle()))
    assert rt.plan_info()["pass1"] == "generic"


def test_two_threads_asking_for_the_same_pattern_compile_it_once(tmp_path):
    """csrc/ssx_jit.h `get`: two contexts upload the same new mesh pattern at the same moment, both asking for the compilation on
    the calling thread (ssx_set_jit mode 1); one of them compiles, the other waits for that code object."""
    import threading
    import custom_scene as cs
    old = os.environ.get("SSX_CACHE_DIR")
    os.environ["SSX_CACHE_DIR"] = str(tmp_path / "cache")
    try:
        lib = _capi.hip_lib()
        counters = lambda: (lambda a, b: (lib.ssx_jit_counters(C.byref(a), C.byref(b)), (a.value, b.value))[1])(C.c_uint64(), C.c_uint64())
        c = cs.CustomScene("cornell-srgb")
        pos, st, m = c.quads[2]
        pos = pos.copy(); pos[3, 1] += 0.25                 # a pattern of this test's own
        c.quads[2] = (pos, st, m)
        orc = c.oracle()
        desc = c.desc(orc)
        rs = [Renderer(Options(scene_name="cornell-srgb", res=(24, 16), spp=3, seed=8, texture="test-img.png", jit_pass1=True)) for _ in range(2)]
        compiled0, hits0 = counters()
        go = threading.Barrier(2)
        errors = []
        def upload(r):
            try:
                go.wait(); r.upload_scene_desc(desc)
            except Exception as e:                          # noqa: BLE001 -- reported below
                errors.append(e)
        threads = [threading.Thread(target=upload, args=(r,)) for r in rs]
        [t.start() for t in threads]; [t.join() for t in threads]
        assert not errors, errors
        assert counters() == (compiled0 + 1, hits0)
        ref = orc.render(24, 16, 3, seed=8)
        for r in rs:
            assert r.jit_status() == (_capi.SSX_JIT_STATE_SPECIALISED, "") and r.plan_info()["kernel"] == "ssx_render_kernel_jit"
            r.render_start(); r.render_wait()
            assert np.array_equal(bits(r.xyza), bits(ref))
    finally:
        if old is None:
            os.environ.pop("SSX_CACHE_DIR", None)
        else:
            os.environ["SSX_CACHE_DIR"] = old


_JIT_CHILD = r"""
import os, sys, time, json
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import ctypes as C
import numpy as np
import custom_scene as cs
from simple_spectral_amd import Options, Renderer, _capi
c = cs.CustomScene("cornell-srgb")
pos, st, m = c.quads[1]; pos = pos.copy(); pos[2, 1] += 0.5; c.quads[1] = (pos, st, m)
orc = c.oracle()
r = Renderer(Options(scene_name="cornell-srgb", res=(40, 32), spp=4, seed=3, texture="test-img.png"))
t = time.time(); r.upload_scene_desc(c.desc(orc)); up = time.time() - t
state, msg = r.jit_status()
a, b = C.c_uint64(), C.c_uint64(); _capi.hip_lib().ssx_jit_counters(C.byref(a), C.byref(b))
r.render_start(); r.render_wait()
same = bool(np.array_equal(r.xyza.view(np.uint32), orc.render(40, 32, 4, seed=3).view(np.uint32)))
print(json.dumps({"state": state, "msg": msg, "compiled": a.value, "disk_hits": b.value, "upload_s": up, "kernel": r.plan_info()["kernel"], "same": same}))
"""


def test_pass1_compiled_in_the_background_and_kept_on_disk(tmp_path):
    """the specialisation of pass 1 to a scene's own mesh topology is ON by default and costs the caller nothing.
    A scene that matches no built-in pattern starts on the generic kernel at once; the compilation runs on a background thread once
    the scene has rendered enough (or is asked for through ssx_jit_status), the context switches kernels at the start of a later
    render -- same bits before and after -- and the code object goes to the disk cache, from which a second PROCESS starts
    specialised without compiling.  A damaged cache file is ignored; a failing compiler leaves the generic kernel in place."""
    import json, subprocess, sys, time
    import torch
    import custom_scene as cs
    cache = tmp_path / "cache"
    old = os.environ.get("SSX_CACHE_DIR")
    os.environ["SSX_CACHE_DIR"] = str(cache)
    try:
        lib = _capi.hip_lib()
        counters = lambda: (lambda a, b: (lib.ssx_jit_counters(C.byref(a), C.byref(b)), (a.value, b.value))[1])(C.c_uint64(), C.c_uint64())
        c = cs.CustomScene("cornell-srgb")
        pos, st, m = c.quads[1]
        pos = pos.copy(); pos[2, 1] += 0.5                  # a pattern no other test compiles (the in-memory cache is per process)
        c.quads[1] = (pos, st, m)
        orc = c.oracle()
        r = Renderer(Options(scene_name="cornell-srgb", res=(40, 32), spp=4, seed=3, texture="test-img.png"))
        compiled0, hits0 = counters()
        t = time.time(); r.upload_scene_desc(c.desc(orc)); upload = time.time() - t
        assert r.jit_status()[0] == _capi.SSX_JIT_STATE_GENERIC_MEANWHILE and r.plan_info()["pass1"] == "generic"
        ref = orc.render(40, 32, 4, seed=3)
        r.render_start(); r.render_wait()
        assert np.array_equal(bits(r.xyza), bits(ref))
        assert counters() == (compiled0, hits0) and not cache.exists()       # 5120 samples: nobody asked the compiler
        # a render large enough asks for it (32 M samples on the generic kernel); it arrives a second or two later
        out = torch.zeros((512, 512, 4), device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        r.render_device(out.data_ptr(), stream, width=512, height=512, spp=128)
        small = torch.zeros((32, 40, 4), device="cuda")
        t0 = time.time()
        while r.jit_status()[0] == _capi.SSX_JIT_STATE_GENERIC_MEANWHILE and time.time() - t0 < 120:
            r.render_device(small.data_ptr(), stream)                          # every render polls; this one also checks the bits
            torch.cuda.synchronize()
            assert np.array_equal(bits(small.cpu().numpy()), bits(ref))
            time.sleep(0.05)
        waited = time.time() - t0
        assert r.jit_status() == (_capi.SSX_JIT_STATE_SPECIALISED, ""), (r.jit_status(), waited)
        assert r.plan_info()["kernel"] == "ssx_render_kernel_jit" and counters() == (compiled0 + 1, hits0)
        r.render_device(small.data_ptr(), stream); torch.cuda.synchronize()
        assert np.array_equal(bits(small.cpu().numpy()), bits(ref))
        files = sorted(cache.glob("pass1-*.co"))
        assert len(files) == 1 and files[0].stat().st_size > 20000
        # a second process: specialised straight from the upload, nothing compiled
        child = _JIT_CHILD % {"root": os.path.dirname(HERE)}
        run = lambda **env: json.loads(subprocess.run([sys.executable, "-c", child], env=dict(os.environ, **env), capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
        second = run()
        assert second["state"] == _capi.SSX_JIT_STATE_SPECIALISED and second["compiled"] == 0 and second["disk_hits"] == 1 and second["same"] and second["kernel"] == "ssx_render_kernel_jit", second
        assert second["upload_s"] < upload + 1.0, (second, upload)          # (both uploads include the calibration render; "compiled == 0" above is the statement, this only says the upload did not sit through a compilation: ~1.5 s; tools/jit_rate.py has the numbers)
        # a damaged file is not trusted
        data = files[0].read_bytes()
        files[0].write_bytes(data[:len(data) // 2])
        third = run()
        assert third["state"] == _capi.SSX_JIT_STATE_GENERIC_MEANWHILE and third["disk_hits"] == 0 and third["same"], third
        files[0].write_bytes(data[:-9] + bytes([data[-9] ^ 1]) + data[-8:])  # one payload bit flipped: the checksum says no
        assert run()["state"] == _capi.SSX_JIT_STATE_GENERIC_MEANWHILE
        # a compiler that fails (here: told to) leaves the scene on the generic kernel, with the reason
        files[0].unlink()
        failing = "import os; os.environ['SSX_DEBUG_ENV'] = '1'; os.environ['SSX_JIT_FAIL'] = '1'\n" + child.replace("state, msg = r.jit_status()", "state, msg = r.jit_status(-1)")
        out = json.loads(subprocess.run([sys.executable, "-c", failing], env=dict(os.environ), capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
        assert out["state"] == _capi.SSX_JIT_STATE_FAILED and "SSX_JIT_FAIL" in out["msg"] and out["same"] and out["kernel"] == "ssx_render_kernel", out
        # ... also when the caller wanted the compilation at upload (this used to fail the upload)
        atup = failing.replace('texture="test-img.png"))', 'texture="test-img.png", jit_pass1=True))')
        out = json.loads(subprocess.run([sys.executable, "-c", atup], env=dict(os.environ), capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
        assert out["state"] == _capi.SSX_JIT_STATE_FAILED and out["same"], out
        # no cache directory at all: the code lives in the process only
        nocache = run(SSX_CACHE_DIR="")
        assert nocache["state"] == _capi.SSX_JIT_STATE_GENERIC_MEANWHILE and not list(cache.glob("pass1-*"))
        # a cache directory that group or others can write to is not trusted with code that runs on the GPU: neither read nor written
        # (ADVICE r04); the same directory closed to them (0700) is used
        shared = tmp_path / "shared"
        shared.mkdir()
        (shared / files[0].name).write_bytes(data)
        os.chmod(shared, 0o777)
        loose = run(SSX_CACHE_DIR=str(shared))
        assert loose["state"] == _capi.SSX_JIT_STATE_GENERIC_MEANWHILE and loose["disk_hits"] == 0 and loose["same"], loose
        os.chmod(shared, 0o700)
        tight = run(SSX_CACHE_DIR=str(shared))
        assert tight["state"] == _capi.SSX_JIT_STATE_SPECIALISED and tight["disk_hits"] == 1 and tight["compiled"] == 0 and tight["same"], tight
    finally:
        if old is None:
            os.environ.pop("SSX_CACHE_DIR", None)
        else:
            os.environ["SSX_CACHE_DIR"] = old


def test_shadow_queue_layouts_wide_by_default_narrow_when_it_buys_a_workgroup():
    """The shadow-ray queues have 48-byte entries (the contribution rides along, the flush writes the finished
    next-event term) unless 32-byte entries (contribution to HBM at park time, visibility byte at flush time)
    let a fourth workgroup live on a CU: the CIE 2006 tables on the Cornell topology.  Both layouts give the same
    bits; with SSX_NARROW_QUEUE set a subprocess repeats the oracle comparisons of the default scenes that way."""
    import subprocess, sys
    r = Renderer(Options(scene_name="cornell-srgb", res=(8, 8), spp=1, texture="test-img.png"))
    r.render_start(); r.render_wait()
    assert r.kernel_info()["max_blocks_per_cu"] == 4 and r.kernel_info()["lds_bytes"] > 32768   # wide
    assert r.plan_info()["kernel"] == "ssx_render_kernel_cornell"
    r6 = Renderer(Options(scene_name="cornell-srgb", res=(8, 8), spp=1, texture="test-img.png", observer=2006))
    assert r6.kernel_info()["max_blocks_per_cu"] == 4                                              # narrow buys the fourth
    assert r6.plan_info()["kernel"] == "ssx_render_kernel_cornell_nq"
    env = dict(os.environ, SSX_DEBUG_ENV="1", SSX_NARROW_QUEUE="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k",
                          "bit_exact_against_oracle or config1 or many_units or without_explicit"], env=env, capture_output=True, text=True, cwd=os.path.dirname(HERE))
    assert out.returncode == 0, out.stdout[-3000:]


def test_rccl_communicators_are_kept_between_combines():
    """ssx_reduce_rccl used to build and destroy its communicators at every call.  They now stay in the
    contexts: a second combine of the same contexts creates none, the image is the one-rank sum (itself) both times, another
    context gets its own group, and destroying a context releases its communicator."""
    lib = _capi.hip_lib()
    lib.ssx_rccl_groups_made.restype = C.c_uint64
    _, r = gpu_render(scene_name="cornell-srgb", res=(32, 24), spp=4, seed=1, texture="test-img.png")
    ref = r.xyza.copy()
    ctxs = (C.c_void_p * 1)(r._ctx)
    before = lib.ssx_rccl_groups_made()
    for _ in range(3):
        assert lib.ssx_reduce_rccl(ctxs, 1, 32, 24) == 0, lib.ssx_last_error(r._ctx)
        out = np.zeros_like(ref)
        assert lib.ssx_read_framebuffer(r._ctx, out.ctypes.data) == 0
        assert np.array_equal(bits(out), bits(ref))
    assert lib.ssx_rccl_groups_made() == before + 1
    _, r2 = gpu_render(scene_name="cornell", res=(32, 24), spp=2)
    assert lib.ssx_reduce_rccl((C.c_void_p * 1)(r2._ctx), 1, 32, 24) == 0
    assert lib.ssx_rccl_groups_made() == before + 2
    r2.close()                                      # releases its communicator; the first context's is untouched
    assert lib.ssx_reduce_rccl(ctxs, 1, 32, 24) == 0 and lib.ssx_rccl_groups_made() == before + 2


def test_render_device_can_be_captured_in_a_hip_graph():
    """ssx_render_device only enqueues (no allocation or device-wide synchronisation on its path once the buffers are
    sized): a render can be captured into a hipGraph and replayed -- the launch-bound small renders (BASELINE configs[0]: five
    kernels and three fills in 0.4 ms) are what that is for.  The replayed image is the oracle's, every time."""
    import torch
    W, H, spp = 128, 128, 16
    r = Renderer(Options(scene_name="cornell-srgb", res=(W, H), spp=spp, seed=3, texture="test-img.png"))
    ref = ol.Oracle("cornell-srgb", texture="test-img.png").render(W, H, spp, seed=3)
    out = torch.zeros((H, W, 4), device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        r.render_device(out.data_ptr(), side.cuda_stream)          # sizes every buffer (and asks the occupancy calculator) outside the capture
        r.render_device_wait()                                     # nothing of this context is queued any more
        assert np.array_equal(bits(out.cpu().numpy()), bits(ref))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            r.render_device(out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(bits(out.cpu().numpy()), bits(ref))
    r.render_device(out.data_ptr(), torch.cuda.current_stream().cuda_stream)   # and the context still renders the plain way
    torch.cuda.synchronize()
    assert np.array_equal(bits(out.cpu().numpy()), bits(ref))
