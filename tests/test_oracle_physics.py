"""The oracle against radiometry, not against a reading of the reference.

SURVEY 8(c) / VERDICT r03: the integrator part of the oracle (intersection, light sampling, recursion) cannot be pinned against
the reference itself here (it needs GLM, which the image lacks), and the reference holds no vector for it.  What CAN be had without
the reference is a check of the restated formulas against closed-form radiometry -- a mis-transcribed pdf, n.l, area or depth
test shows as a bias here, whatever both sides of the parity tests agree on:

  * a furnace: a closed box whose walls all emit E and reflect rho (flat spectra) has L = E (1 + rho + ... + rho^9) under the
    reference's depth limit (MAX_DEPTH 10, src/stdafx.hpp:47; src/renderer.cpp:166-250), with explicit light sampling
    (emission seen by the camera only, direct light by next-event estimation over the twelve wall triangles) and without it
    (emission at every hit) -- two estimators, one number;
  * Lambert's formula: the irradiance a polygonal emitter sends to a point is (E/2) sum_i beta_i n.Gamma_i over its edges; a
    floor patch under a square light, black elsewhere, has exactly L = rho/pi times that (the bounce finds no light to count).

Both are ratios against a render whose radiance is E by construction (the camera looks at an emitter of albedo 0), sample by
sample with the same wavelengths, so the colour pipeline cancels.  CPU only; Monte-Carlo tolerances are four standard errors,
estimated from the pixel means.  TEST INFRASTRUCTURE (this is a test of oracle/, not of the product)."""
import numpy as np

import custom_scene as cs


def _flat_materials(c, rho, emit):
    """a Lambertian material with flat albedo rho and flat emission emit over 300..900 nm"""
    a = c.add_spectrum([rho, rho], 300.0, 900.0)
    e = c.add_spectrum([emit, emit], 300.0, 900.0)
    return c.add_material(kind=0, albedo_spectrum=a, emission_spectrum=e)


def _box(c, m, lo=-100.0, hi=100.0):
    # (large on purpose: the reference rejects hits nearer than EPS = 0.001, src/geometry.cpp:88 with src/stdafx.hpp:58, so rays that start
    # within a millimetre of an edge of the box leave it; in a box of 200 units that is one path in ~10^5, in a box of 2 one in 500)
    l, h = lo, hi                                       # normals inward (src/scene.cpp:372-409 vertex order)
    c.add_quad((l, l, h), (l, l, l), (l, h, l), (l, h, h), m)
    c.add_quad((h, l, l), (h, l, h), (h, h, h), (h, h, l), m)
    c.add_quad((l, l, h), (h, l, h), (h, l, l), (l, l, l), m)
    c.add_quad((h, h, h), (l, h, h), (l, h, l), (h, h, l), m)
    c.add_quad((l, l, l), (h, l, l), (h, h, l), (l, h, l), m)
    c.add_quad((h, l, h), (l, l, h), (l, h, h), (h, h, h), m)


def _furnace(rho, observer=1931):
    c = cs.CustomScene("cornell", observer=observer, keep_quads=False)
    _box(c, _flat_materials(c, rho, 1.0))
    c.set_camera((30.0, -20.0, 10.0), (-100.0, 40.0, 90.0), up=(0, 1, 0), vfov_deg=70.0)
    return c.oracle()


def _mean_and_se(ratio_pixels):
    v = ratio_pixels.reshape(-1)
    return float(v.mean()), float(v.std(ddof=1) / np.sqrt(v.size))


def test_furnace_radiance_with_and_without_light_sampling():
    W = H = 64
    spp = 16
    unit = _furnace(0.0).render(W, H, spp, seed=5)                      # L = E for every sample: the yardstick, per pixel
    assert (unit[..., 3] == 1.0).all() and (unit[..., 1] > 0).all()      # closed: every camera ray hits
    for rho in (0.5, 0.8):
        want = sum(rho ** k for k in range(10))                          # E (1 - rho^10) / (1 - rho)
        o = _furnace(rho)
        for els in (True, False):
            img = o.render(W, H, spp, seed=5, els=els)
            assert (img[..., 3] == 1.0).all() and np.isfinite(img).all()
            for ch in range(3):                                          # X, Y, Z: the same factor in each
                mean, se = _mean_and_se(img[..., ch].astype(np.float64) / unit[..., ch].astype(np.float64))
                assert se < 0.005 * want, (rho, els, ch, se)
                assert abs(mean - want) < 4.0 * se + 1e-5 * want, (rho, els, ch, mean, want, se)
    # without light sampling and with rho = 0.5 the estimator has no variance at all beyond float rounding: every path collects
    # E at each of its ten hits, weighted rho^depth (cos / pi over the cosine pdf cancels up to rounding)
    img = _furnace(0.5).render(W, H, spp, seed=5, els=False)
    r = img[..., 1].astype(np.float64) / unit[..., 1].astype(np.float64)
    assert (np.abs(r - sum(0.5 ** k for k in range(10))) < 2e-5).mean() > 0.999       # (all but the odd path that left through the EPS band)


def _lambert_irradiance(p, n, poly):
    """(1/2) sum_i beta_i n.Gamma_i for a polygon of unit radiance seen from p (vertices counter-clockwise as seen from p)"""
    v = [np.asarray(q, dtype=np.float64) - p for q in poly]
    v = [q / np.linalg.norm(q) for q in v]
    total = 0.0
    for i in range(len(v)):
        a, b = v[i], v[(i + 1) % len(v)]
        g = np.cross(a, b)
        total += np.arccos(np.clip(a @ b, -1.0, 1.0)) * (n @ (g / np.linalg.norm(g)))
    return 0.5 * abs(total)


def test_direct_light_of_a_square_emitter_follows_lamberts_formula():
    rho = 0.6
    light = [(-0.5, 1.0, -0.25), (0.75, 1.0, -0.25), (0.75, 1.0, 1.0), (-0.5, 1.0, 1.0)]      # y = 1, off-centre, facing down
    for target in ((0.0, 0.0, 0.0), (0.9, 0.0, -0.6), (-1.2, 0.0, 0.4)):
        c = cs.CustomScene("cornell", keep_quads=False)
        floor = _flat_materials(c, rho, 0.0)
        lamp = _flat_materials(c, 0.0, 1.0)
        c.add_quad((-3, 0, 3), (3, 0, 3), (3, 0, -3), (-3, 0, -3), floor)                       # normal +y
        c.add_quad(light[0], light[1], light[2], light[3], lamp)                                # normal -y
        eye = np.array(target) + np.array((0.7, 0.9, -0.8))
        c.set_camera(tuple(eye), target, up=(0, 1, 0), vfov_deg=0.05)                           # a pixel footprint of ~1e-3: one point
        o = c.oracle()
        img = o.render(16, 16, 64, seed=2)
        # the yardstick: the same camera rays (same seed, same wavelengths per sample) onto an emitter of albedo 0
        u = cs.CustomScene("cornell", keep_quads=False)
        u.add_quad((-3, 0, 3), (3, 0, 3), (3, 0, -3), (-3, 0, -3), _flat_materials(u, 0.0, 1.0))
        u.set_camera(tuple(eye), target, up=(0, 1, 0), vfov_deg=0.05)
        unit = u.oracle().render(16, 16, 64, seed=2)
        assert (img[..., 3] == 1.0).all() and (unit[..., 3] == 1.0).all()
        want = rho / np.pi * _lambert_irradiance(np.array(target, dtype=np.float64), np.array((0.0, 1.0, 0.0)), light)
        assert want > 0.01
        mean, se = _mean_and_se(img[..., 1].astype(np.float64) / unit[..., 1].astype(np.float64))
        assert se < 0.02 * want, (target, se, want)
        assert abs(mean - want) < 4.0 * se + 1e-4 * want, (target, mean, want, se)


def _cornell_triangles():
    """the triangles of the built-in Cornell box as the HOST library lays them out (quad q -> 2q: v00 v10 v11, 2q + 1: v00 v11 v01;
    src/geometry.cpp:128-139), float64"""
    from simple_spectral_amd.renderer import Scene
    scene = Scene("cornell")                            # (kept alive while its description is read)
    d = scene.desc.contents
    tris = []
    for q in range(d.n_quads):
        Q = d.quads[q]
        v = [np.array(x.pos[:], dtype=np.float64) for x in (Q.v00, Q.v10, Q.v11, Q.v01)]
        tris.append((q, v[0], v[1], v[2]))
        tris.append((q, v[0], v[2], v[3]))
    return tris


def test_closest_hit_against_moeller_trumbore_in_binary64():
    """orc_scene_intersect (the reference's watertight test in binary32 with a binary64 fallback, src/geometry.cpp:12-101, and the closest-hit
    loop, src/scene.cpp:433-445) against the textbook Moeller-Trumbore test in binary64 on 20000 random rays through the Cornell box:
    same primitive, same distance (to binary32 accuracy), same plane normal, for every ray that is not within 1e-5 of an edge or of the EPS cut."""
    import ctypes as C
    import oracle_lib as ol
    o = ol.Oracle("cornell")
    tris = _cornell_triangles()
    A = np.array([t[1] for t in tris]); E1 = np.array([t[2] - t[1] for t in tris]); E2 = np.array([t[3] - t[1] for t in tris])
    prim = np.array([t[0] for t in tris])
    g = np.random.default_rng(11)
    lo, hi = A.min(axis=0), (A + np.maximum(E1, E2)).max(axis=0)
    checked = misses = 0
    for _ in range(20000):
        orig = g.uniform(lo + 0.05, hi - 0.05).astype(np.float32)
        d = g.normal(size=3); d = (d / np.linalg.norm(d)).astype(np.float32)
        O, D = orig.astype(np.float64), d.astype(np.float64)
        P = np.cross(D, E2); det = (E1 * P).sum(axis=1)
        ok = np.abs(det) > 1e-12
        inv = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0)
        T = O - A
        u = (T * P).sum(axis=1) * inv
        Qv = np.cross(T, E1)
        v = (Qv * D).sum(axis=1) * inv
        t = (E2 * Qv).sum(axis=1) * inv
        w = 1.0 - u - v
        edge = np.minimum(np.minimum(u, v), w)
        inside = ok & (edge > 0) & (t >= 1e-3)
        marginal = ok & (np.abs(edge) < 1e-5) & (t > 0) | ok & (edge > -1e-5) & (np.abs(t - 1e-3) < 1e-5)
        if marginal.any():
            continue
        ray = ol.Ray(ol.V3(*map(float, orig)), ol.V3(*map(float, d)))
        hit = ol.Hit()
        got = o.lib.orc_scene_intersect(o.scene, C.byref(ray), C.byref(hit), -1, None)
        if not inside.any():
            assert not got, (orig, d)
            misses += 1
            continue
        ts = np.where(inside, t, np.inf)
        k = int(np.argmin(ts))
        if np.sort(ts)[1] - ts[k] < 1e-5 * max(1.0, ts[k]):        # two surfaces at one distance: list order decides, not geometry
            continue
        assert got and hit.prim == prim[k], (orig, d, hit.prim, prim[k])
        assert abs(hit.dist - ts[k]) <= 1e-5 * ts[k] + 5e-5, (hit.dist, ts[k])   # binary32 arithmetic on coordinates of size ~5: a few 1e-5 absolute
        n = np.cross(E1[k], E2[k]); n /= np.linalg.norm(n)
        hn = np.array((hit.normal.x, hit.normal.y, hit.normal.z), dtype=np.float64)
        assert abs(abs(hn @ n) - 1.0) < 1e-5                         # the triangle's plane normal
        checked += 1
    assert checked > 15000, (checked, misses)


def _solid_angle(p, a, b, c):
    """Van Oosterom & Strackee: the solid angle of triangle abc seen from p"""
    ra, rb, rc = a - p, b - p, c - p
    la, lb, lc = np.linalg.norm(ra), np.linalg.norm(rb), np.linalg.norm(rc)
    num = ra @ np.cross(rb, rc)
    den = la * lb * lc + (ra @ rb) * lc + (ra @ rc) * lb + (rb @ rc) * la
    return abs(2.0 * np.arctan2(num, den))


def test_light_sampling_is_uniform_in_solid_angle_with_the_stated_pdf():
    """Scene::get_rand_toward_light (src/scene.cpp:417-431) -> PrimQuad::get_rand_toward (src/geometry.cpp:141-145) -> Arvo's sampler
    (src/util/spherical-tri.cpp, src/util/random.cpp:108-151): a uniformly chosen light, one of its two triangles, a direction uniform in that
    triangle's solid angle, pdf = 1 / (2 Omega_triangle n_lights).  Checked without the sampler's formulas: 1/pdf equals twice the
    Van Oosterom-Strackee solid angle of one half of the chosen light, the direction lies in that half, and the fraction of directions falling into a
    sub-triangle equals the ratio of solid angles (uniformity)."""
    import ctypes as C
    import oracle_lib as ol
    o = ol.Oracle("cornell")
    lib = o.lib
    lib.orc_scene_get_rand_toward_light.restype = None
    lib.orc_scene_get_rand_toward_light.argtypes = [C.c_void_p, C.POINTER(ol.Rng), ol.V3, C.POINTER(ol.V3), C.POINTER(C.c_int), C.POINTER(C.c_float)]
    tris = _cornell_triangles()
    n_prims, n_lights, n_mat = C.c_int(), C.c_int(), C.c_int()
    lib.orc_scene_counts(o.scene, C.byref(n_prims), C.byref(n_lights), C.byref(n_mat))
    lights = [lib.orc_scene_light(o.scene, i) for i in range(n_lights.value)]
    assert len(lights) >= 1
    rng = ol.Rng(); lib.orc_seed_sample(1, 2, 3, C.byref(rng))
    for p in ((0.5, -2.0, 0.3), (-2.5, 0.0, 2.0), (2.0, 1.5, -1.0)):
        P = np.array(p, dtype=np.float64)
        N = 6000
        in_sub = np.zeros(2 * len(tris)); total = np.zeros(2 * len(tris))
        for _ in range(N):
            d, light, pdf = ol.V3(), C.c_int(), C.c_float()
            lib.orc_scene_get_rand_toward_light(o.scene, C.byref(rng), ol.V3(*p), C.byref(d), C.byref(light), C.byref(pdf))
            assert light.value in lights
            D = np.array((d.x, d.y, d.z), dtype=np.float64)
            assert abs(np.linalg.norm(D) - 1.0) < 1e-5
            # which half was sampled: the one whose solid angle (Van Oosterom-Strackee, binary64) the pdf states
            halves = (2 * light.value, 2 * light.value + 1)
            omegas = [_solid_angle(P, *tris[k][1:]) for k in halves]
            stated = 1.0 / pdf.value / (2.0 * len(lights))
            k = halves[int(np.argmin([abs(stated - w) for w in omegas]))]
            omega = omegas[halves.index(k)]
            assert abs(stated - omega) < 2e-4 * omega, (k, pdf.value, omegas)     # (alpha + beta + gamma - pi in binary32: ~1e-5 relative here)
            # ... and the direction lies in that half (binary64 Moeller-Trumbore; the binary32 sampler may land 1e-4 outside an edge)
            _, a, b, c = tris[k]
            e1, e2 = b - a, c - a
            pv = np.cross(D, e2); det = e1 @ pv
            tv = P - a; u = (tv @ pv) / det
            qv = np.cross(tv, e1); v = (D @ qv) / det
            assert (e2 @ qv) / det > 0 and min(u, v, 1.0 - u - v) > -3e-4, (p, k, u, v)
            total[k] += 1
            # the sub-triangle (a, midpoint ab, midpoint ac) = {u + v < 1/2}
            in_sub[k] += (u + v) < 0.5
        for k in np.nonzero(total)[0]:
            _, a, b, c = tris[k]
            frac = _solid_angle(P, a, 0.5 * (a + b), 0.5 * (a + c)) / _solid_angle(P, a, b, c)
            se = np.sqrt(frac * (1 - frac) / total[k])
            assert abs(in_sub[k] / total[k] - frac) < 4.5 * se, (p, k, in_sub[k] / total[k], frac, total[k])
        # the two halves of a light are chosen with equal probability (src/geometry.cpp:141-145), the lights uniformly
        for l in lights:
            n0, n1 = total[2 * l], total[2 * l + 1]
            assert abs(n0 - n1) < 4.5 * np.sqrt(n0 + n1)


def test_flat_radiance_gives_the_integrals_of_the_colour_matching_functions():
    """Hero-wavelength sampling (src/renderer.cpp:138), the table lookups (src/spectrum.cpp:39-67) and flux -> XYZ (src/renderer.cpp:266-273
    with the 0.001 / 1000 scaling of :292-298): for radiance 1 at every wavelength the expected pixel is (int xbar, int ybar, int zbar) d lambda
    -- the trapezoid sums of the observer's table, computed here from the data file, not through the oracle."""
    import os
    data = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")
    for observer, name, step in ((1931, "cie1931-xyzbar-380+5+780.csv", 5.0), (2006, "cie2006-xyzbar-390+1+830.csv", 1.0)):
        tab = np.loadtxt(os.path.join(data, name), delimiter=",")
        want = step * (tab.sum(axis=0) - 0.5 * (tab[0] + tab[-1]))
        img = _furnace(0.0, observer).render(64, 64, 64, seed=5).astype(np.float64)
        px = img[..., :3].reshape(-1, 3)
        mean, se = px.mean(axis=0), px.std(axis=0, ddof=1) / np.sqrt(len(px))
        assert (se < 2e-3 * want).all(), (observer, se)
        assert (np.abs(mean - want) < 4.0 * se + 1e-5 * want).all(), (observer, mean, want, se)


def test_cosine_hemisphere_sampler_and_the_local_frame():
    """rand_coshemi (src/util/random.cpp:33-41, :128-151) and get_rotated_to (src/util/math-helpers.hpp): unit vectors in the upper hemisphere,
    pdf = cos(theta) / pi, moments of the cosine density about the local y axis -- the reference's 'up' -- (E[y] = 2/3, E[y^2] = 1/2, E[x] = E[z] = 0,
    E[x^2] = E[z^2] = 1/4), and the rotation takes the local y axis onto the normal keeping lengths and the angle to it."""
    import ctypes as C
    import oracle_lib as ol
    lib = ol.load()
    rng = ol.Rng(); lib.orc_seed_sample(3, 1, 4, C.byref(rng))
    N = 40000
    w = np.zeros((N, 3)); pdf = np.zeros(N)
    g = np.random.default_rng(2)
    for i in range(N):
        p = C.c_float()
        v = lib.orc_rand_coshemi(C.byref(rng), C.byref(p))
        w[i] = (v.x, v.y, v.z); pdf[i] = p.value
        if i < 2000:
            n = g.normal(size=3); n = (n / np.linalg.norm(n)).astype(np.float32)
            r = lib.orc_get_rotated_to(v, ol.V3(*map(float, n)))
            R = np.array((r.x, r.y, r.z))
            assert abs(np.linalg.norm(R) - 1.0) < 1e-5 and abs(R @ n.astype(np.float64) - v.y) < 1e-5, (n, w[i], R)
    assert np.abs(np.linalg.norm(w, axis=1) - 1.0).max() < 1e-5 and (w[:, 1] >= 0).all()
    assert np.abs(pdf - w[:, 1] / np.pi).max() < 1e-6
    for got, want, sd in ((w[:, 1].mean(), 2 / 3, np.sqrt(1 / 18)), ((w[:, 1] ** 2).mean(), 0.5, np.sqrt(1 / 12)),
                          (w[:, 0].mean(), 0.0, 0.5), (w[:, 2].mean(), 0.0, 0.5), ((w[:, 0] ** 2).mean(), 0.25, 0.25), ((w[:, 2] ** 2).mean(), 0.25, 0.25)):
        assert abs(got - want) < 4.5 * sd / np.sqrt(N), (got, want)
