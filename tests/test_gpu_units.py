"""Per-function and crafted-scene parity (VERDICT r01 "closable
parity gaps"): the HIP building blocks, run through the C ABI's ssx_debug_eval / ssx_debug_samples,
against the oracle's unit-level functions and per-sample results -- bit for bit, on inputs that
provably reach the rare branches (tests/test_unit_cases_cpu.py holds the branch-counter proofs).
Run with -m gpu on MI355X."""
import ctypes as C
import os

import numpy as np
import pytest

import crafted
import custom_scene as cs
import oracle_lib as ol
import unit_cases as uc
from unit_cases import unit
from simple_spectral_amd import Options, Renderer, _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def same_bits_or_both_nan(got_u32, ref_u32, float_cols):
    """uint32 [n, k] arrays; the float columns may differ in NaN payload only"""
    g, r = np.asarray(got_u32), np.asarray(ref_u32)
    ok = g == r
    for c in float_cols:
        ok[:, c] |= np.isnan(g[:, c].view(np.float32)) & np.isnan(r[:, c].view(np.float32))
    return ok


@pytest.fixture(scope="module")
def cornell():
    return Renderer(Options(scene_name="cornell-srgb", res=(8, 8), spp=1, texture="test-img.png")), ol.Oracle("cornell-srgb", texture="test-img.png")


def custom_pair(c):
    orc = c.oracle()
    r = Renderer(Options(scene_name="cornell", res=(8, 8), spp=1, observer=c.observer))
    r.upload_scene_desc(c.desc(orc))
    return r, orc


def test_fmath_functions_equal_the_header_on_the_host(cornell):
    r, orc = cornell
    w = uc.fmath_inputs()
    got = r.debug_eval(_capi.SSX_DBG_FMATH, w, 5)
    ref = bits(uc.oracle_fmath(orc.lib, w))
    assert same_bits_or_both_nan(got, ref, range(5)).all()


def test_spherical_triangle_incl_degenerate_ladder(cornell):
    """src/util/spherical-tri.cpp:18-124 on random, tiny, coinciding, antipodal, coplanar and NaN vertices"""
    r, orc = cornell
    tri = uc.sphtri_inputs()
    got = r.debug_eval(_capi.SSX_DBG_SPHTRI, uc.f2u(tri), 5)
    ref = bits(uc.oracle_sphtri(orc.lib, tri))
    ok = same_bits_or_both_nan(got, ref, range(5))
    assert ok.all(), (np.argwhere(~ok)[:5], tri[np.argwhere(~ok)[0][0]])


def test_arvo_sampler_incl_denominator_zero(cornell):
    """src/util/random.cpp:101-154"""
    r, orc = cornell
    w = uc.arvo_inputs(orc.lib)
    got = r.debug_eval(_capi.SSX_DBG_ARVO, w, 5)
    ref = uc.oracle_arvo(orc.lib, w)
    ok = same_bits_or_both_nan(got, ref, range(3))
    assert ok.all(), np.argwhere(~ok)[:5]


def test_cosine_hemisphere_incl_rejection_retry(cornell):
    """src/util/random.cpp:29-49 + math-helpers.hpp:14-39"""
    r, orc = cornell
    w = uc.coshemi_inputs()
    assert int(uc.coshemi_draws(w).sum()) >= 40
    got = r.debug_eval(_capi.SSX_DBG_COSHEMI, w, 6)
    ref = uc.oracle_coshemi(orc.lib, w)
    assert np.array_equal(got, ref)


def test_rand_choice_incl_lemire_redraw(cornell):
    """src/util/random.hpp:75-78 -> libstdc++ uniform_int_distribution (Lemire)"""
    r, orc = cornell
    w = uc.rand_choice_inputs()
    assert int(uc.lemire_redraws(w).sum()) > 300
    assert np.array_equal(r.debug_eval(_capi.SSX_DBG_RAND_CHOICE, w, 3), uc.oracle_rand_choice(orc.lib, w))
    w = uc.rng_words(3000, 11)
    assert np.array_equal(r.debug_eval(_capi.SSX_DBG_RAND_1F, w, 3), uc.oracle_rand_1f(orc.lib, w))


@pytest.mark.parametrize("observer", [1931, 2006])
def test_flux_to_xyz(observer):
    """src/util/color.hpp:115-139"""
    r = Renderer(Options(scene_name="cornell", res=(8, 8), spp=1, observer=observer))
    orc = ol.Oracle("cornell", observer=observer)
    d = r.scene.desc.contents
    w = uc.flux_inputs(d.lambda_min, d.lambda_step)
    assert np.array_equal(r.debug_eval(_capi.SSX_DBG_FLUX_TO_XYZ, w, 3), bits(uc.oracle_flux(orc, w)))


def scene_quads(c):
    return [q[0] for q in c.quads]


@pytest.mark.parametrize("which", ["cornell-srgb", "shared-edge", "degenerate"])
def test_scene_intersect_on_vertices_edges_and_random_rays(which):
    """src/scene.cpp:433-445, src/geometry.cpp:12-139: closest quad, distance and st, incl. rays aimed
    exactly at vertices / edge midpoints / the diagonal, axis-parallel rays, rays leaving a quad"""
    c = {"cornell-srgb": lambda: cs.CustomScene("cornell-srgb"), "shared-edge": crafted.shared_edge_scene,
         "degenerate": lambda: crafted.degenerate_light_scene("room")}[which]()
    r, orc = custom_pair(c)
    w = uc.trace_inputs(scene_quads(c))
    got = r.debug_eval(_capi.SSX_DBG_TRACE, w, 5)
    ref, st = uc.oracle_trace(orc, w)
    assert st.tri_f64 > (200 if which != "cornell-srgb" else 20)          # the f64 fallback ran (oracle counter)
    hit = ref[:, 0] != 0xFFFFFFFF
    assert hit.mean() > 0.5
    assert np.array_equal(got[:, 0], ref[:, 0])                            # same quad (or both none)
    assert np.array_equal(got[hit][:, 2:5], ref[hit][:, 2:5])               # same dist and st, bit for bit


@pytest.mark.parametrize("view", ["room", "edge_ab"])
def test_light_sampling_from_degenerate_positions(view):
    """src/scene.cpp:417-431 -> geometry.cpp:103-145: shading points on the extension of a light edge, on a
    light vertex (NaN directions), in the light's plane, far from a tiny light, and random ones"""
    c = crafted.degenerate_light_scene(view)
    r, orc = custom_pair(c)
    g = np.random.default_rng(5)
    e = 2e-3
    pts = np.concatenate([
        g.uniform(-3.9, 3.9, size=(2000, 3)),
        np.array([3, 1, 0]) + g.uniform(-e, e, size=(600, 3)) * [0, 1, 1],    # beyond v10 on the line v00-v10
        np.array([1, 1, 3]) + g.uniform(-e, e, size=(600, 3)) * [1, 1, 0],    # beyond v11 on the line v10-v11
        np.array([[0, 1, 0], [1, 1, 0], [1, 1, 1], [0, 1, 1], [0.5, 1, 0.5], [0.5, 1, 0], [2, 1, 2], [-3, 1, 0.5]], dtype=np.float64),  # on vertices / in the light's plane
        np.stack([g.uniform(-3, 3, 300), np.full(300, 1.0), g.uniform(-3, 3, 300)], axis=1),
    ]).astype(np.float32)
    w = uc.sample_light_inputs(pts)
    st = ol.Stats()
    orc.lib.orc_debug_set_stats(C.byref(st))
    try:
        ref = uc.oracle_sample_light(orc, w)
    finally:
        orc.lib.orc_debug_set_stats(None)
    assert st.sphtri_half_pi > 50 and st.sphtri_only_a > 50 and st.sphtri_nan > 500 and st.light_pdf_inf > 500
    got = r.debug_eval(_capi.SSX_DBG_SAMPLE_LIGHT, w, 7)
    ok = same_bits_or_both_nan(got, ref, (0, 1, 2, 4))
    assert ok.all(), (np.argwhere(~ok)[:5], pts[np.argwhere(~ok)[0][0]])


@pytest.mark.parametrize("view", ["edge_ab", "edge_bc", "far", "room"])
def test_degenerate_light_scene_per_sample(view):
    """Whole paths through the degenerate ladder, zero-area light triangles (pdf = +inf) and a collapsed
    triangle: every SAMPLE (XYZA and draws consumed) equals the oracle's, then the image."""
    c = crafted.degenerate_light_scene(view)
    r, orc = custom_pair(c)
    W, H, spp = 24, 24, 4
    r.options.res = (W, H); r.options.spp = spp; r.options.seed = 1
    xyza, state, levels = r.debug_samples()
    ref_xyza, ref_state, st = orc.samples(W, H, spp, seed=1)
    assert st.sphtri_nan > 50 and st.light_pdf_inf > 50
    assert np.array_equal(state, ref_state)                  # same number of draws, sample by sample
    assert np.array_equal(bits(xyza), bits(ref_xyza))
    r.xyza = np.zeros((H, W, 4), dtype=np.float32)
    r.render_start(); r.render_wait()
    assert np.array_equal(bits(r.xyza), bits(orc.render(W, H, spp, seed=1)))


def test_shared_edge_scene_per_sample():
    """Camera rays a few ulps around a vertex shared by four quads (src/geometry.cpp:56-67 f64 fallback; ties)."""
    c = crafted.shared_edge_scene()
    r, orc = custom_pair(c)
    W, H, spp = 24, 24, 4
    r.options.res = (W, H); r.options.spp = spp; r.options.seed = 1
    xyza, state, _ = r.debug_samples()
    ref_xyza, ref_state, st = orc.samples(W, H, spp, seed=1)
    assert st.tri_f64 > 2000
    assert np.array_equal(state, ref_state) and np.array_equal(bits(xyza), bits(ref_xyza))


@pytest.mark.parametrize("scene,observer,io,els", [("cornell-srgb", 1931, False, True), ("cornell", 2006, False, True), ("plane-srgb", 1931, False, True),
                                                   ("cornell-srgb", 1931, True, True), ("cornell-srgb", 1931, False, False)])
def test_per_sample_xyza_and_draws(scene, observer, io, els):
    """per sample, not only pixel means -- 4608 samples per case: XYZA bits, the PCG32
    state after the last draw (= draws consumed) and the number of continued levels."""
    tex = None if scene == "cornell" else "test-img.png"
    W, H, spp = 24, 16, 12
    r = Renderer(Options(scene_name=scene, observer=observer, res=(W, H), spp=spp, seed=3, texture=tex, indirect_only=io, explicit_light_sampling=els))
    xyza, state, levels = r.debug_samples()
    orc = ol.Oracle(scene, observer=observer, texture=tex)
    if not els and scene == "plane-srgb":
        orc.lib.orc_scene_set_material_kind(orc.scene, orc.lib.orc_scene_quad_material(orc.scene, 0), 1)
    ref_xyza, ref_state, st = orc.samples(W, H, spp, seed=3, indirect_only=io, els=els)
    assert np.array_equal(state, ref_state)
    assert np.array_equal(bits(xyza), bits(ref_xyza))
    # levels = interactions that continued; the oracle's histogram counts interactions per sample
    hist = np.bincount(levels.ravel(), minlength=10)
    assert hist.sum() == W * H * spp
    assert int((levels.astype(np.int64)).sum()) <= st.interactions


@pytest.mark.parametrize("op,name", [(_capi.SSX_SWEEP_SIN_PROOF, "sin"), (_capi.SSX_SWEEP_COS_PROOF, "cos"), (_capi.SSX_SWEEP_ACOS_PROOF, "acos")])
def test_fmath_header_proved_against_an_independent_evaluation(cornell, op, name):
    """include/ssx_fmath.h is shared by the oracle and the kernels, so their bit-equality says
    nothing about the header itself, and its only outside check was ~50 k mpmath inputs.  Here EVERY ONE of the 2^32 float
    patterns goes through ssx_sinf / ssx_cosf / ssx_acosf and through csrc/ssx_ddmath.h -- double-double Taylor series,
    three-part pi/2, Newton on the cosine: nothing shared with the header; pinned against mpmath by tests/test_fmath.py -- on the
    device: the header returns the correctly rounded value (NaN outside its domain) for every input but the five listed below (and
    ssx_sincosf, which the samplers call, the same two floats as ssx_sinf and ssx_cosf for every input),
    and the inputs the independent evaluation cannot decide (within 2^-70 of a rounding boundary), if any, are settled with mpmath."""
    import mpmath as mp
    import oracle_lib as ol
    r, _ = cornell
    bad, undecided, examples = r.debug_sweep(op)
    wrong = sorted(e for e in examples if not (e >> 32))
    # The header evaluates in binary64 (error ~2^-52) and rounds once, so a value closer than that to a float rounding boundary can
    # fall on the wrong side: include/ssx_fmath.h says so; THIS is the complete list (and sin(-0) = +0, the sign of a zero).
    known = {"sin": [0x46199998, 0x80000000, 0xC6199998], "cos": [], "acos": [0x328885A3, 0x39826222]}[name]
    assert bad == len(known) and wrong == known, (name, bad, [hex(e) for e in wrong])
    assert undecided <= 8 - len(known), (name, undecided)       # all of them are in the example list
    lib = ol.load()
    mp.mp.prec = 300
    fn, ref = {"sin": (lib.orc_sinf, mp.sin), "cos": (lib.orc_cosf, mp.cos), "acos": (lib.orc_acosf, mp.acos)}[name]
    for e in examples:
        x = float(np.array([e & 0xFFFFFFFF], dtype=np.uint32).view(np.float32)[0])
        v = ref(mp.mpf(x))
        f = np.float32(float(v))
        want = min((np.nextafter(f, np.float32(-np.inf)), f, np.nextafter(f, np.float32(np.inf))), key=lambda c: abs(mp.mpf(float(c)) - v))
        got = np.float32(fn(x))
        if e >> 32:                     # the independent evaluation could not decide: mpmath does
            assert got == want, (name, hex(e))
        elif x == 0.0:
            assert got == 0.0 and not np.signbit(got)       # sin(-0) = +0 where the correctly rounded sine keeps the sign
        else:                           # a known exception: the neighbour of the correctly rounded value, the true value within 2^-52 of the boundary
            assert got != want and abs(int(got.view(np.int32)) - int(want.view(np.int32))) == 1, (name, hex(e))
            mid = (mp.mpf(float(got)) + mp.mpf(float(want))) / 2
            assert abs(v - mid) < abs(v) * mp.mpf(2) ** -52, (name, hex(e))
    print("%s: 2^32 inputs, %d known exceptions, %d decided by mpmath" % (name, len(known), undecided))


@pytest.mark.parametrize("op,name", [
    (_capi.SSX_SWEEP_RCP, "1.0f/x"), (_capi.SSX_SWEEP_SQRT, "sqrt"), (_capi.SSX_SWEEP_INVERSESQRT, "inversesqrt"),
    (_capi.SSX_SWEEP_SIN, "sin"), (_capi.SSX_SWEEP_COS, "cos"), (_capi.SSX_SWEEP_ACOS, "acos"), (_capi.SSX_SWEEP_DIV_PI, "x/pi"),
    (_capi.SSX_SWEEP_RCP64, "binary64 reciprocal"), (_capi.SSX_SWEEP_DIV_PAIRS, "division pairs"),
    (_capi.SSX_SWEEP_ACOS_SIN, "fused acos + sin")])
def test_exhaustive_sweeps_of_the_cheaper_exact_arithmetic(cornell, op, name):
    """Every one of the 2^32 float inputs, on the device: the kernel's cheaper forms of 1/x, sqrt, 1/sqrt,
    x/pi, sin, cos, acos against the operation / the include/ssx_fmath.h function that DEFINES the result
    (simple_spectral_amd/csrc/ssx_exact.h states why they are exact; this is the check), the binary64
    reciprocal behind the shared-divisor divisions (error <= 1 ulp for every float divisor, which the
    exactness argument needs), and 3 x 2^32 hashed (numerator, divisor) pairs through that division."""
    r, _ = cornell
    bad, mx, examples = r.debug_sweep(op)
    if op in (_capi.SSX_SWEEP_SQRT, _capi.SSX_SWEEP_INVERSESQRT, _capi.SSX_SWEEP_RCP) and bad:
        # specified domains (ssx_exact.h): sqrt_normal for x >= 2^-100, +-0, +inf, NaN and negative normal x;
        # rcp for 2^-126 <= |x| <= 2^126 (normal input, normal result), +-0, +-inf, NaN
        f32 = lambda v: int(np.array([v], dtype=np.float32).view(np.uint32)[0])
        if op == _capi.SSX_SWEEP_RCP:
            parts = [(f32(2.0 ** -126), f32(2.0 ** 126)), (f32(-2.0 ** -126), f32(-2.0 ** 126)), (0, 0), (0x80000000, 0x80000000), (0x7F800000, 0x7FFFFFFF), (0xFF800000, 0xFFFFFFFF)]
        else:
            parts = [(f32(2.0 ** -100), 0x7FFFFFFF), (0, 0), (0x80000000, 0x80000000), (f32(-2.0 ** -126), 0xFFFFFFFF)]
        report = []
        for lo, hi in parts:
            b, _, ex = r.debug_sweep(op, lo=lo, count=hi - lo + 1)
            report.append((hex(lo), hex(hi), b, [hex(e) for e in ex[:4]]))
        assert all(p[2] == 0 for p in report), (name, report)
        print("%s: exact on its specified domain; %d inputs outside it differ, e.g. %s" % (name, bad, [hex(e) for e in examples[:4]]))
        return
    assert bad == 0, (name, bad, [hex(e) for e in examples])
    if op == _capi.SSX_SWEEP_RCP64:
        assert mx <= 1
    if op == _capi.SSX_SWEEP_ACOS_SIN:
        # 2 * 0x3F800000 + 2 inputs in [-1, 1]; the rounding test / the |x| ~ 1 threshold send a small share to ssx_sinf_lds
        n = 2 * 0x3F800000 + 2
        print("fused acos+sin: %d of %d inputs fall back (%.3f %%)" % (mx, n, 100.0 * mx / n))
        assert mx < 0.05 * n


def test_triangle_primitives_and_triangle_lights():
    """PrimTri as a primitive and as a light (src/geometry.hpp:55-74, src/geometry.cpp:103-116 against
    :141-145).  Light sampling from random points: direction, light, pdf and the stream position -- a triangle light consumes
    one random number fewer than a quad light; intersection incl. the region a quad's second triangle would cover; then every
    sample of a render (XYZA, draws) and the image."""
    c = crafted.triangle_scene()
    r, orc = custom_pair(c)
    assert r.plan_info()["pass1"] == "generic"
    g = np.random.default_rng(12)
    w = uc.sample_light_inputs(g.uniform(-3.9, 3.9, size=(3000, 3)).astype(np.float32))
    ref = uc.oracle_sample_light(orc, w)
    got = r.debug_eval(_capi.SSX_DBG_SAMPLE_LIGHT, w, 7)
    assert same_bits_or_both_nan(got, ref, (0, 1, 2, 4)).all()
    assert set(np.unique(ref[:, 3])) == {6, 7}                                      # both lights picked
    w = uc.trace_inputs(scene_quads(c))
    got = r.debug_eval(_capi.SSX_DBG_TRACE, w, 5)
    ref, _ = uc.oracle_trace(orc, w)
    hit = ref[:, 0] != 0xFFFFFFFF
    assert np.array_equal(got[:, 0], ref[:, 0]) and np.array_equal(got[hit][:, 2:5], ref[hit][:, 2:5])
    assert (ref[:, 0] == 8).sum() > 20 and (ref[:, 0] == 9).sum() > 20 and (ref[:, 0] == 6).sum() > 20
    W, H, spp = 24, 24, 6
    r.options.res = (W, H); r.options.spp = spp; r.options.seed = 4
    xyza, state, _ = r.debug_samples()
    ref_xyza, ref_state, st = orc.samples(W, H, spp, seed=4)
    assert np.array_equal(state, ref_state) and np.array_equal(bits(xyza), bits(ref_xyza))
    r.xyza = np.zeros((H, W, 4), dtype=np.float32)
    r.render_start(); r.render_wait()
    ref = orc.render(W, H, spp, seed=4)
    assert np.array_equal(bits(r.xyza), bits(ref)) and ref[..., :3].max() > 0


@pytest.mark.parametrize("n_prims,observer", [(70, 1931), (128, 2006), (33, 1931)])
def test_more_than_32_primitives(n_prims, observer):
    """Scene::intersect is a loop over any number of primitives (src/scene.cpp:433-445); the kernel works
    through them in groups of 32 in list order.  Quads and triangles in all orientations, lights in several groups.  70
    primitives / CIE 1931: every table in LDS; 128 / CIE 2006: the permuted vertex table is read from HBM."""
    c = crafted.many_prims_scene(n_prims, observer)
    r, orc = custom_pair(c)
    w = uc.trace_inputs(scene_quads(c), n_random=1500)
    got = r.debug_eval(_capi.SSX_DBG_TRACE, w, 5)
    ref, _ = uc.oracle_trace(orc, w)
    hit = ref[:, 0] != 0xFFFFFFFF
    assert np.array_equal(got[:, 0], ref[:, 0]) and np.array_equal(got[hit][:, 2:5], ref[hit][:, 2:5])
    assert len(np.unique(ref[hit][:, 0])) > n_prims * 0.8                           # nearly every primitive is some ray's closest hit
    W, H, spp = 32, 24, 4
    r.options.res = (W, H); r.options.spp = spp; r.options.seed = 6
    xyza, state, _ = r.debug_samples()
    ref_xyza, ref_state, st = orc.samples(W, H, spp, seed=6)
    assert np.array_equal(state, ref_state) and np.array_equal(bits(xyza), bits(ref_xyza))
    r.xyza = np.zeros((H, W, 4), dtype=np.float32)
    r.render_start(); r.render_wait()
    ref = orc.render(W, H, spp, seed=6)
    assert np.array_equal(bits(r.xyza), bits(ref)) and ref[..., :3].max() > 0


def test_many_textures():
    """Nine textures of different sizes (1 x 1, non-square, 256 x 2) on the walls, the floor and the short block: each
    Lambertian material holds its own texture (src/material.cpp:10-29); the reference's scenes use one."""
    c = crafted.many_textures_scene(9)
    r, orc = custom_pair(c)
    W, H, spp = 40, 40, 4
    r.options.res = (W, H); r.options.spp = spp; r.options.seed = 12
    xyza, state, _ = r.debug_samples()
    ref_xyza, ref_state, st = orc.samples(W, H, spp, seed=12)
    assert np.array_equal(state, ref_state) and np.array_equal(bits(xyza), bits(ref_xyza))
    r.xyza = np.zeros((H, W, 4), dtype=np.float32)
    r.render_start(); r.render_wait()
    ref = orc.render(W, H, spp, seed=12)
    assert np.array_equal(bits(r.xyza), bits(ref)) and ref[..., :3].max() > 0


@pytest.mark.parametrize("seed", range(48))
def test_random_scenes(seed):
    """Differential fuzz: a scene, a camera and the render switches drawn from the seed (tests/crafted.py random_scene), rendered
    by both sides -- per sample (XYZA and the final PCG32 state, i.e. the draws consumed) and as an image."""
    c, o = crafted.random_scene(seed)
    r, orc = custom_pair(c)
    W, H, spp = 24, 20, 3
    r.options.res = (W, H); r.options.spp = spp; r.options.seed = seed
    r.options.indirect_only = o["indirect_only"]; r.options.explicit_light_sampling = o["els"]; r.options.flat_field_correction = o["flat_field"]
    xyza, state, _ = r.debug_samples()
    ref_xyza, ref_state, st = orc.samples(W, H, spp, seed=seed, indirect_only=o["indirect_only"], els=o["els"], flat_field=o["flat_field"])
    assert np.array_equal(state, ref_state)
    assert np.array_equal(bits(xyza), bits(ref_xyza)) or same_bits_or_both_nan(bits(xyza).reshape(-1, 4), bits(ref_xyza).reshape(-1, 4), range(4)).all()
    r.xyza = np.zeros((H, W, 4), dtype=np.float32)
    r.render_start(); r.render_wait()
    ref = orc.render(W, H, spp, seed=seed, indirect_only=o["indirect_only"], els=o["els"], flat_field=o["flat_field"])
    assert np.array_equal(bits(r.xyza), bits(ref)) or same_bits_or_both_nan(bits(r.xyza).reshape(-1, 4), bits(ref).reshape(-1, 4), range(4)).all()


@pytest.mark.parametrize("seed", range(36))
def test_warped_builtin_scenes_on_their_topology_kernels(seed):
    """The fuzz of the kernels the headline numbers are measured on (tests/crafted.py warped_builtin): the Cornell box and the plane scene with every
    distinct corner moved the same way wherever it occurs -- rotated / scaled / translated, jittered, flattened into slivers -- so the library keeps the
    kernel whose pass 1 is specialised to the mesh topology (positions are run-time data there), against the oracle per sample and as an image."""
    c, base, o = crafted.warped_builtin(seed)
    orc = c.oracle()
    r = Renderer(Options(scene_name=base, res=(8, 8), spp=1, texture=None if base == "cornell" else "test-img.png", observer=c.observer))
    r.upload_scene_desc(c.desc(orc))
    forced_generic = os.environ.get("SSX_DEBUG_ENV") == "1" and os.environ.get("SSX_GENERIC_KERNEL", "0")[:1] not in ("", "0")   # (tools/test_kernel_variants.sh)
    assert r.plan_info()["pass1"] == ("generic" if forced_generic else "plane topology" if base == "plane-srgb" else "cornell topology")
    W, H, spp = 28, 20, 4
    r.options.res = (W, H); r.options.spp = spp; r.options.seed = 300 + seed
    r.options.indirect_only = o["indirect_only"]; r.options.explicit_light_sampling = o["els"]; r.options.flat_field_correction = o["flat_field"]
    xyza, state, _ = r.debug_samples()
    ref_xyza, ref_state, st = orc.samples(W, H, spp, seed=300 + seed, indirect_only=o["indirect_only"], els=o["els"], flat_field=o["flat_field"])
    assert np.array_equal(state, ref_state)
    assert np.array_equal(bits(xyza), bits(ref_xyza)) or same_bits_or_both_nan(bits(xyza).reshape(-1, 4), bits(ref_xyza).reshape(-1, 4), range(4)).all()
    r.xyza = np.zeros((H, W, 4), dtype=np.float32)
    r.render_start(); r.render_wait()
    ref = orc.render(W, H, spp, seed=300 + seed, indirect_only=o["indirect_only"], els=o["els"], flat_field=o["flat_field"])
    assert np.array_equal(bits(r.xyza), bits(ref)) or same_bits_or_both_nan(bits(r.xyza).reshape(-1, 4), bits(ref).reshape(-1, 4), range(4)).all()


def test_scene_limits_are_errors_not_surprises():
    c = crafted.many_prims_scene(129)
    orc = c.oracle()
    r = Renderer(Options(scene_name="cornell", res=(8, 8), spp=1))
    with pytest.raises(Exception) as e:
        r.upload_scene_desc(c.desc(orc))
    assert "n_quads" in str(e.value)
    c = crafted.triangle_scene()
    pos, st, m = c.quads[8]
    pos = pos.copy(); pos[1, 0] = 3e9                                                # beyond 2^30
    c.quads[8] = (pos, st, m)
    with pytest.raises(Exception) as e:
        r.upload_scene_desc(c.desc(c.oracle()))
    assert "2^30" in str(e.value)
    c = crafted.many_textures_scene(65)
    with pytest.raises(Exception) as e:
        r.upload_scene_desc(c.desc(c.oracle()))
    assert "too many textures" in str(e.value)


def test_light_sampling_a_hair_from_a_light_vertex_at_the_origin():
    """ssx_exact::sqrt_normal is exact for x >= 2^-100 only; squared distances below that (possible only next to the
    origin) must take the plain IEEE sequence.  Shading points 1e-16 ... 1e-30 from a light vertex at (0, 0, 0): direction, pdf
    and stream equal the oracle's, bit for bit (NaN where the reference produces NaN)."""
    c = crafted.origin_light_scene()
    r, orc = custom_pair(c)
    g = np.random.default_rng(21)
    mags = 10.0 ** g.uniform(-30, -14, size=(3000, 1))
    pts = (unit(g.normal(size=(3000, 3))) * mags).astype(np.float32)
    pts[:50, 1] = 0.0                                        # in the light's plane
    pts[50:100] = np.abs(pts[50:100])                        # inside the light's corner
    assert (np.sum(pts.astype(np.float64) ** 2, axis=1) < 2.0 ** -100).sum() > 1500
    w = uc.sample_light_inputs(pts)
    ref = uc.oracle_sample_light(orc, w)
    got = r.debug_eval(_capi.SSX_DBG_SAMPLE_LIGHT, w, 7)
    ok = same_bits_or_both_nan(got, ref, (0, 1, 2, 4))
    assert ok.all(), (np.argwhere(~ok)[:5], pts[np.argwhere(~ok)[0][0]])


_BLACK_CHILD = r"""
import os, sys, json, hashlib
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
from simple_spectral_amd import Options, Renderer
out = {}
for io, els in ((False, True), (True, True), (False, False)):
    r = Renderer(Options(scene_name="plane-srgb", res=(24, 16), spp=12, seed=3, texture="test-img.png", indirect_only=io, explicit_light_sampling=els))
    xyza, state, levels = r.debug_samples()
    out["%%d%%d" %% (io, els)] = hashlib.sha256(xyza.tobytes() + state.tobytes() + levels.tobytes()).hexdigest()
print(json.dumps(out))
"""

# the Cornell box with a black floor and black texels on the generic kernel (SSX_GENERIC_KERNEL=1 in the child's environment): the kernels of the
# Cornell topology are compiled without the shortcut (path_step<NARROW, BLACK = false>), the generic one has it -- lanes of both kinds in one wave
_BLACK_CORNELL_CHILD = r"""
import os, sys, json, hashlib
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import custom_scene as cs
from simple_spectral_amd import Options, Renderer
c = cs.CustomScene("cornell-srgb")
zero = c.add_spectrum(np.zeros(8, dtype=np.float32), 400.0, 700.0)
black = c.add_material(kind=0, albedo_spectrum=zero)
pos, st_, _m = c.quads[0]
c.quads[0] = (pos, st_, black)
c.textures[0][::2, :, :] = 0
orc = c.oracle()
r = Renderer(Options(scene_name="cornell-srgb", res=(8, 8), spp=1, texture="test-img.png"))
r.upload_scene_desc(c.desc(orc))
assert r.plan_info()["pass1"] == "generic", r.plan_info()
r.options.res = (24, 24); r.options.spp = 8; r.options.seed = 4
xyza, state, levels = r.debug_samples()
print(json.dumps({"sha": hashlib.sha256(xyza.tobytes() + state.tobytes() + levels.tobytes()).hexdigest()}))
"""


def test_black_surfaces_end_the_path_on_the_random_draws_alone():
    """Round 6: a black Lambertian surface (plane-srgb's light box, src/scene.cpp:357-413: albedo 0) ends its path in path_step without the
    light sampler's and the hemisphere sampler's arithmetic -- whatever they produce is multiplied by f_s = 0 -- but WITH their random
    draws (csrc/ssx_kernels.hip).  Per sample against the oracle, which evaluates everything: XYZA bits, the final PCG32 state (= the draws
    consumed, rejection loops included) and the levels, for plane-srgb with and without --indirect-only and without explicit light sampling
    (the plane is a mirror then: src/scene.cpp:346-355), in RGB mode, and for a Cornell box whose floor is black and whose red wall carries
    a texture with black texels; a light whose emission is not finite switches the shortcut off (header flag), same bits; and the
    library with the shortcut disabled (SSX_BLACK_SHORTCUT=0) returns the same per-sample arrays."""
    import hashlib, json, subprocess, sys
    import custom_scene as cs
    got = {}
    for io, els in ((False, True), (True, True), (False, False)):
        r = Renderer(Options(scene_name="plane-srgb", res=(24, 16), spp=12, seed=3, texture="test-img.png", indirect_only=io, explicit_light_sampling=els))
        xyza, state, levels = r.debug_samples()
        orc = ol.Oracle("plane-srgb", texture="test-img.png")
        if not els:
            orc.lib.orc_scene_set_material_kind(orc.scene, orc.lib.orc_scene_quad_material(orc.scene, 0), 1)
        ref_xyza, ref_state, st = orc.samples(24, 16, 12, seed=3, indirect_only=io, els=els)
        assert np.array_equal(state, ref_state) and np.array_equal(bits(xyza), bits(ref_xyza)), (io, els)
        if els and not io:
            assert st.path_len_hist[2] > 0.9 * 24 * 16 * 12              # nearly every path's second hit is a black wall of the light box (S = 2): the shortcut's case
        got["%d%d" % (io, els)] = hashlib.sha256(xyza.tobytes() + state.tobytes() + levels.tobytes()).hexdigest()
    # the same library with the shortcut off
    env = dict(os.environ, SSX_DEBUG_ENV="1", SSX_BLACK_SHORTCUT="0")
    out = subprocess.run([sys.executable, "-c", _BLACK_CHILD % {"root": ROOT}], env=env, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    assert json.loads(out.stdout.strip().splitlines()[-1]) == got
    # RGB mode: albedo {r, g, b, 0}
    r = Renderer(Options(scene_name="plane-srgb", res=(16, 16), spp=6, seed=9, texture="test-img.png", render_mode="rgb"))
    xyza, state, _ = r.debug_samples()
    ref_xyza, ref_state, _st = ol.Oracle("plane-srgb", texture="test-img.png", rgb=True).samples(16, 16, 6, seed=9)
    assert np.array_equal(state, ref_state) and np.array_equal(bits(xyza), bits(ref_xyza))
    # a Cornell box with a black floor (constant albedo 0) and black texels on the textured wall: lanes of both kinds in one wave
    c = cs.CustomScene("cornell-srgb")
    zero = c.add_spectrum(np.zeros(8, dtype=np.float32), 400.0, 700.0)
    black = c.add_material(kind=0, albedo_spectrum=zero)
    pos, st_, _m = c.quads[0]
    c.quads[0] = (pos, st_, black)
    c.textures[0][::2, :, :] = 0                                       # every other texel row black
    orc = c.oracle()
    r = Renderer(Options(scene_name="cornell-srgb", res=(8, 8), spp=1, texture="test-img.png"))
    r.upload_scene_desc(c.desc(orc))
    r.options.res = (24, 24); r.options.spp = 8; r.options.seed = 4
    xyza, state, levels = r.debug_samples()
    ref_xyza, ref_state, st = orc.samples(24, 24, 8, seed=4)
    assert np.array_equal(state, ref_state) and np.array_equal(bits(xyza), bits(ref_xyza))
    assert st.interactions > 0
    # ... which the Cornell topology's kernel evaluates in full (compiled without the shortcut: profiles/r06/NOTES.md section 8); the generic kernel,
    # which takes the shortcut on the lanes that sit on the floor or a black texel, returns the same per-sample arrays
    env = dict(os.environ, SSX_DEBUG_ENV="1", SSX_GENERIC_KERNEL="1")
    out = subprocess.run([sys.executable, "-c", _BLACK_CORNELL_CHILD % {"root": ROOT}], env=env, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    assert json.loads(out.stdout.strip().splitlines()[-1])["sha"] == hashlib.sha256(xyza.tobytes() + state.tobytes() + levels.tobytes()).hexdigest()
    # an emission table that is not "finite and below 2^60": the flag is off, everything is evaluated (same bits as the oracle either way)
    c2 = cs.CustomScene("plane-srgb")
    li = next(i for i, m in enumerate(c2.materials) if c2.spectra[m["emission_spectrum"]][0].any())
    data, low, high = c2.spectra[c2.materials[li]["emission_spectrum"]]
    data = data.copy(); data[3] = np.float32(2.0 ** 61)
    c2.spectra[c2.materials[li]["emission_spectrum"]] = (data, low, high)
    orc2 = c2.oracle()
    r2 = Renderer(Options(scene_name="plane-srgb", res=(8, 8), spp=1, texture="test-img.png"))
    r2.upload_scene_desc(c2.desc(orc2))
    r2.options.res = (16, 16); r2.options.spp = 4; r2.options.seed = 2
    xyza, state, _ = r2.debug_samples()
    ref_xyza, ref_state, _st = orc2.samples(16, 16, 4, seed=2)
    assert np.array_equal(state, ref_state) and np.array_equal(bits(xyza), bits(ref_xyza))


def test_image_wider_than_65535_tiles_plane_kernel_reads_its_samples():
    """The plane-topology kernels make a sample where a lane takes it, from the tile's column and row the work unit carries in 16 bits each
    (csrc/ssx_kernels.hip WorkUnit::txy); an image of more than 65 535 tiles in a direction keeps the generate kernel (csrc/ssx_api.hip
    enqueue_front).  One pixel row of 65 537 tiles, ragged at the end, against the oracle."""
    W, H, spp = 8 * 65536 + 3, 1, 1
    r = Renderer(Options(scene_name="plane-srgb", res=(W, H), spp=spp, seed=5, texture="test-img.png"))
    r.render_start(); r.render_wait()
    ref = ol.Oracle("plane-srgb", texture="test-img.png").render(W, H, spp, seed=5)
    assert np.array_equal(bits(r.xyza), bits(ref))
    # and the neighbour below the limit (65 535 tiles: the fused path), same check
    W2 = 8 * 65535 - 2
    r2 = Renderer(Options(scene_name="plane-srgb", res=(W2, 1), spp=1, seed=5, texture="test-img.png"))
    r2.render_start(); r2.render_wait()
    ref2 = ol.Oracle("plane-srgb", texture="test-img.png").render(W2, 1, 1, seed=5)
    assert np.array_equal(bits(r2.xyza), bits(ref2))
