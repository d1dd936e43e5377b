"""CPU side of the per-function and crafted-scene parity tests: the inputs really reach the rare
branches (oracle branch counters), and the custom-scene path of the oracle equals its built-in
scenes.  The GPU side (tests/test_gpu_units.py) compares the HIP functions with these results."""
import ctypes as C

import numpy as np
import pytest

import crafted
import custom_scene as cs
import oracle_lib as ol
import unit_cases as uc


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("scene", ["cornell", "cornell-srgb", "plane-srgb"])
def test_custom_scene_description_reproduces_the_builtin_scene(scene):
    c = cs.CustomScene(scene)
    o = c.oracle()
    assert np.array_equal(bits(o.render(40, 32, 3, seed=5)), bits(ol.Oracle(scene).render(40, 32, 3, seed=5)))
    d = c.desc(o)
    assert d.n_quads == (7 if scene == "plane-srgb" else 19) and d.n_lights == (6 if scene == "plane-srgb" else 1)


@pytest.mark.parametrize("view,need", [
    ("edge_ab", ("sphtri_half_pi", "sphtri_nan", "light_pdf_inf", "arvo_sin_alpha_le0", "tri_f64")),
    ("edge_bc", ("sphtri_only_a", "sphtri_nan", "light_pdf_inf", "tri_f64")),
    ("far", ("sphtri_nan", "light_pdf_inf")),
    ("room", ("sphtri_regular", "sphtri_half_pi", "sphtri_only_a", "sphtri_nan", "light_pdf_inf", "nee_visible")),
])
def test_degenerate_light_scene_reaches_the_ladder(view, need):
    """src/util/spherical-tri.cpp:74-123, src/geometry.cpp:115"""
    o = crafted.degenerate_light_scene(view).oracle()
    img, st = o.render(24, 24, 4, seed=1, stats=True)
    d = st.as_dict()
    for k in need:
        assert d[k] > 50, (k, d[k])
    assert np.isfinite(img).all()  # the NaN directions never pass `n_dot_l > 0` (src/renderer.cpp:193)


def test_shared_edge_scene_reaches_the_f64_fallback():
    """src/geometry.cpp:56-67"""
    o = crafted.shared_edge_scene().oracle()
    img, st = o.render(24, 24, 4, seed=1, stats=True)
    assert st.tri_f64 > 2000 and st.hits == st.samples


def test_unit_inputs_cover_the_branches():
    lib = ol.load()
    st = ol.Stats()
    lib.orc_debug_set_stats(C.byref(st))
    try:
        tri = uc.sphtri_inputs()
        uc.oracle_sphtri(lib, tri)
        assert st.sphtri_regular > 3000 and st.sphtri_half_pi > 100 and st.sphtri_only_a > 100 and st.sphtri_nan > 100
        words = uc.arvo_inputs(lib)
        uc.oracle_arvo(lib, words)
        assert st.arvo_denom_zero >= 4 and st.arvo_sin_alpha_le0 > 100 and st.funcbar_zero > 10
        w = uc.coshemi_inputs()
        before = st.coshemi_retries
        uc.oracle_coshemi(lib, w)
        assert st.coshemi_retries - before >= 40 == int(uc.coshemi_draws(w).sum())
        w = uc.rand_choice_inputs()
        before = st.lemire_redraws
        uc.oracle_rand_choice(lib, w)
        assert st.lemire_redraws - before >= int(uc.lemire_redraws(w).sum()) > 300
    finally:
        lib.orc_debug_set_stats(None)


def test_triangle_light_draws_one_number_fewer_and_has_one_triangle():
    """PrimTri::get_rand_toward (src/geometry.cpp:103-116) against PrimQuad::get_rand_toward (:141-145), PrimTri::intersect
    against PrimQuad::intersect (:128-139): what the oracle does with the triangle kind of tests/crafted.py."""
    c = crafted.triangle_scene()
    o = c.oracle()
    w = uc.sample_light_inputs(np.random.default_rng(1).uniform(-3, 3, size=(400, 3)).astype(np.float32))
    out = uc.oracle_sample_light(o, w)
    # number of PCG32 steps from the input state to the output state
    def steps(i):
        st = int(w[i, 3]) | (int(w[i, 4]) << 32); inc = int(w[i, 5]) | (int(w[i, 6]) << 32)
        end = int(out[i, 5]) | (int(out[i, 6]) << 32)
        for n in range(12):
            if st == end:
                return n
            st = (st * 6364136223846793005 + inc) & 0xFFFFFFFFFFFFFFFF
        return -1
    n_tri = {steps(i) for i in range(len(w)) if out[i, 3] == 6}
    n_quad = {steps(i) for i in range(len(w)) if out[i, 3] == 7}
    assert n_tri == {3} and n_quad == {4}            # light pick + [triangle pick] + two for the spherical triangle
    # a ray through the point that a QUAD of the triangle's vertices would cover with its second triangle passes primitive 8
    ray = ol.Ray(ol.V3(-2.0, 3.0, 1.8), ol.V3(0.0, -1.0, 0.0)); hit = ol.Hit()
    assert o.lib.orc_scene_intersect(o.scene, C.byref(ray), C.byref(hit), -1, None) and hit.prim != 8
    ray = ol.Ray(ol.V3(-1.0, 3.0, 1.0), ol.V3(0.0, -1.0, 0.0))
    assert o.lib.orc_scene_intersect(o.scene, C.byref(ray), C.byref(hit), -1, None) and hit.prim == 8


def test_many_prims_scene_shapes():
    for n, obs in ((33, 1931), (70, 1931), (128, 2006)):
        c = crafted.many_prims_scene(n, obs)
        assert len(c.quads) == n and sum(1 for k in c.kinds.values() if k == "tri") > n // 8
        o = c.oracle()
        img, st = o.render(16, 12, 2, seed=6, stats=True)
        assert np.isfinite(img).all() and img[..., :3].max() > 0 and st.nee_visible > 50


def test_many_textures_scene_shapes():
    c = crafted.many_textures_scene(9)
    assert len(c.textures) == 9 and len({t.shape for t in c.textures}) == 9
    used = {c.materials[m]["albedo_texture"] for _, _, m in c.quads if c.materials[m]["albedo_mode"] == 1}
    assert used == set(range(9))                                                     # every texture is some quad's albedo
    o = c.oracle()
    img, st = o.render(16, 16, 2, seed=12, stats=True)
    assert np.isfinite(img).all() and img[..., :3].max() > 0


def test_random_scenes_are_valid_and_varied():
    kinds, seen_opts = set(), set()
    for seed in range(12):
        c, opts = crafted.random_scene(seed)
        assert 3 <= len(c.quads) <= 116 and any(m == crafted.LIGHT for _, _, m in c.quads)
        kinds |= {c.materials[m]["kind"] for _, _, m in c.quads}
        if c.kinds: kinds.add("tri")
        if any(c.materials[m]["albedo_mode"] for _, _, m in c.quads): kinds.add("tex")
        seen_opts.add(tuple(sorted(opts.items())))
        img = c.oracle().render(12, 10, 2, seed=seed, indirect_only=opts["indirect_only"], els=opts["els"], flat_field=opts["flat_field"])
        assert img.shape == (10, 12, 4)
    assert kinds >= {0, 1, "tri", "tex"} and len(seen_opts) >= 4


def test_warped_builtin_scenes_keep_the_sharing_pattern():
    """tests/crafted.py warped_builtin: the corners of the warped scene coincide exactly where the base scene's do (what the library's choice of the
    topology-specialised kernel looks at), every kind of warp and every base scene occurs, and the oracle renders them."""
    import custom_scene as cs
    seen = set()
    for seed in range(12):
        c, base, opts = crafted.warped_builtin(seed)
        b = cs.CustomScene(base)
        ids_w, ids_b = {}, {}
        pat_w = [[ids_w.setdefault(np.asarray(p, np.float32).tobytes(), len(ids_w)) for p in pos] for pos, _, _ in c.quads]
        pat_b = [[ids_b.setdefault(np.asarray(p, np.float32).tobytes(), len(ids_b)) for p in pos] for pos, _, _ in b.quads]
        assert pat_w == pat_b and len(ids_w) == (12 if base == "plane-srgb" else 28)
        assert any(not np.array_equal(pw, pb) for (pw, _, _), (pb, _, _) in zip(c.quads, b.quads))
        seen.add((base, (seed // 3) % 4))
        img, st = c.oracle().render(12, 10, 2, seed=seed, indirect_only=opts["indirect_only"], els=opts["els"], flat_field=opts["flat_field"], stats=True)
        assert img.shape == (10, 12, 4) and st.interactions > 0
    assert len(seen) == 12
