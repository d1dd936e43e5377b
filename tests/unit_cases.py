"""Inputs for the per-function parity tests and the oracle side of
them: each builder returns the [n, words] input array of one ssx_debug_eval op, each `oracle_*`
function evaluates the oracle's unit-level function on the same rows.  TEST INFRASTRUCTURE."""
import ctypes as C

import numpy as np

import oracle_lib as ol

U32 = np.uint32


def f2u(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def u2f(a):
    return np.ascontiguousarray(a, dtype=np.uint32).view(np.float32)


def rng_words(n, seed):
    """n PCG32 streams {state lo, state hi, inc lo, inc hi} (inc odd)."""
    g = np.random.default_rng(seed)
    w = g.integers(0, 2 ** 32, size=(n, 4), dtype=np.uint64).astype(np.uint32)
    w[:, 2] |= 1
    return w


def pcg32_outputs(words, count):
    """first `count` outputs of each stream, vectorised (src/util/random.hpp:52-58)."""
    state = (words[:, 1].astype(np.uint64) << np.uint64(32)) | words[:, 0].astype(np.uint64)
    inc = (words[:, 3].astype(np.uint64) << np.uint64(32)) | words[:, 2].astype(np.uint64)
    out = np.zeros((len(words), count), dtype=np.uint32)
    with np.errstate(over="ignore"):
        for k in range(count):
            xs = (((state >> np.uint64(18)) ^ state) >> np.uint64(27)).astype(np.uint32)
            rot = (state >> np.uint64(59)).astype(np.uint32)
            out[:, k] = (xs >> rot) | (xs << ((np.uint32(0) - rot) & np.uint32(31)))
            state = state * np.uint64(6364136223846793005) + inc
    return out


def unit(v):
    v = np.asarray(v, dtype=np.float64)
    return (v / np.linalg.norm(v, axis=-1, keepdims=True)).astype(np.float32)


# ------------------------------------------------------------------------------- SphericalTriangle ----
def sphtri_inputs(seed=1, n_random=4000):
    """Unit-vector triples: random ones, and the degenerate families of src/util/spherical-tri.cpp:74-123
    (two or three coinciding directions, antipodal pairs, coplanar triples, NaN vertices)."""
    g = np.random.default_rng(seed)
    rows = [unit(g.normal(size=(n_random, 3, 3))).reshape(-1, 9)]
    # small triangles around a random direction (angles 1e-1 .. 1e-7): sides whose cosine rounds to 1
    base = unit(g.normal(size=(1500, 1, 3)))
    scale = 10.0 ** g.uniform(-7, -1, size=(1500, 1, 1))
    rows.append(unit(base + scale * g.normal(size=(1500, 3, 3))).reshape(-1, 9))
    a = unit(g.normal(size=(600, 3)))
    b = unit(g.normal(size=(600, 3)))
    c = unit(g.normal(size=(600, 3)))
    for tri in ((a, a, c), (a, b, a), (a, b, b), (a, a, a), (a, -a, c), (a, b, -b), (a, b, -a), (a, -a, a)):
        rows.append(np.concatenate(tri, axis=1))
    # coplanar with the origin (great-circle triangles: area 0, regular branch with angles 0 / pi)
    t = g.uniform(0, 2 * np.pi, size=(600, 3))
    rows.append(np.stack([np.cos(t), np.sin(t), np.zeros_like(t)], axis=-1).astype(np.float32).reshape(-1, 9))
    # exact axis triples and NaN vertices (normalize(0) of a shading point on a light vertex)
    ax = np.eye(3, dtype=np.float32)
    rows.append(np.array([np.concatenate([ax[i], ax[j], ax[k]]) for i in range(3) for j in range(3) for k in range(3)], dtype=np.float32))
    nanv = np.full(3, np.nan, dtype=np.float32)
    rows.append(np.array([np.concatenate(t3) for t3 in ((nanv, ax[0], ax[1]), (ax[0], nanv, ax[1]), (ax[0], ax[1], nanv), (nanv, nanv, nanv))], dtype=np.float32))
    return np.ascontiguousarray(np.concatenate(rows, axis=0), dtype=np.float32)


def oracle_sphtri(lib, rows):
    out = np.zeros((len(rows), 5), dtype=np.float32)
    classes = np.zeros(len(rows), dtype=np.int32)  # 0 regular, 1 ladder
    t = ol.SphTri()
    for i, r in enumerate(rows):
        lib.orc_sphtri_make(ol.V3(*r[0:3]), ol.V3(*r[3:6]), ol.V3(*r[6:9]), C.byref(t))
        out[i] = (t.b, t.cos_c, t.alpha, t.cos_alpha, t.surface_area)
    return out


def sphtri_full(lib, rows):
    res = []
    t = ol.SphTri()
    for r in rows:
        lib.orc_sphtri_make(ol.V3(*r[0:3]), ol.V3(*r[3:6]), ol.V3(*r[6:9]), C.byref(t))
        res.append((t.b, t.cos_c, t.alpha, t.cos_alpha, t.surface_area))
    return np.array(res, dtype=np.float32)


# ------------------------------------------------------------------------------------------ Arvo ----
def arvo_inputs(lib, seed=2, n=3000):
    """rows = A, B, C, b, cos_c, alpha, cos_alpha, area, rng[4]: spherical triangles made by the oracle
    from sphtri_inputs (so every degenerate family is included), plus hand-set structs that reach
    `denom == 0` (src/util/random.cpp:126-131: u = t - cos_alpha = 0 and v = s + sin_alpha*cos_c = 0)."""
    tri = sphtri_inputs(seed, n_random=n)
    st = sphtri_full(lib, tri)
    rows = np.concatenate([tri, st], axis=1).astype(np.float32)
    # denom == 0: area = 0 -> phi = -alpha, s = sin(-alpha) = -sin(alpha), t = cos(alpha); cos_alpha := t, cos_c := 1
    extra = []
    for alpha in (0.3, 0.7, 1.1, 1.5, 2.0, 2.9):
        al = np.float32(alpha)
        t_ = np.float32(lib.orc_cosf(C.c_float(float(-al))))
        A, B, Cv = unit([1, 0.2, 0.1]), unit([0.1, 1, 0.3]), unit([0.2, 0.1, 1])
        extra.append(np.concatenate([A, B, Cv, [np.float32(0.9), np.float32(1.0), al, t_, np.float32(0.0)]]).astype(np.float32))
    rows = np.concatenate([rows, np.array(extra, dtype=np.float32)], axis=0)
    return np.ascontiguousarray(np.concatenate([f2u(rows), rng_words(len(rows), seed + 100)], axis=1))


def oracle_arvo(lib, words):
    fl = u2f(words[:, :14])
    out = np.zeros((len(words), 5), dtype=np.uint32)
    st = ol.Stats()
    for i in range(len(words)):
        t = ol.SphTri()
        r = fl[i]
        t.A = ol.V3(*r[0:3]); t.B = ol.V3(*r[3:6]); t.C = ol.V3(*r[6:9])
        t.b, t.cos_c, t.alpha, t.cos_alpha, t.surface_area = (float(x) for x in r[9:14])
        rng = ol.Rng(int(words[i, 14]) | (int(words[i, 15]) << 32), int(words[i, 16]) | (int(words[i, 17]) << 32))
        d = lib.orc_rand_toward_sphericaltri(C.byref(rng), C.byref(t))
        out[i, :3] = f2u([d.x, d.y, d.z])
        out[i, 3] = rng.state & 0xFFFFFFFF; out[i, 4] = rng.state >> 32
    return out


# ---------------------------------------------------------------------------------- light sampling ----
def sample_light_inputs(points, seed=3):
    pts = np.ascontiguousarray(points, dtype=np.float32)
    return np.ascontiguousarray(np.concatenate([f2u(pts), rng_words(len(pts), seed)], axis=1))


def oracle_sample_light(orc, words):
    lib = orc.lib
    lib.orc_scene_get_rand_toward_light.argtypes = [C.c_void_p, C.POINTER(ol.Rng), ol.V3, C.POINTER(ol.V3), C.POINTER(C.c_int), C.POINTER(C.c_float)]
    lib.orc_scene_get_rand_toward_light.restype = None
    fl = u2f(words[:, :3])
    out = np.zeros((len(words), 7), dtype=np.uint32)
    d, light, pdf = ol.V3(), C.c_int(), C.c_float()
    for i in range(len(words)):
        rng = ol.Rng(int(words[i, 3]) | (int(words[i, 4]) << 32), int(words[i, 5]) | (int(words[i, 6]) << 32))
        lib.orc_scene_get_rand_toward_light(orc.scene, C.byref(rng), ol.V3(*fl[i]), C.byref(d), C.byref(light), C.byref(pdf))
        out[i, :3] = f2u([d.x, d.y, d.z]); out[i, 3] = light.value; out[i, 4] = f2u([pdf.value])[0]
        out[i, 5] = rng.state & 0xFFFFFFFF; out[i, 6] = rng.state >> 32
    return out


# ---------------------------------------------------------------------------------------- coshemi ----
def coshemi_inputs(seed=4, n=3000, n_retry=40):
    """normal + stream; n_retry streams are searched so that the rejection loop of src/util/random.cpp:29-49
    runs again (second draw of the sample so close to 1 that sqrt(1 - radius_sq) <= EPS)."""
    g = np.random.default_rng(seed)
    normals = unit(g.normal(size=(n + n_retry, 3)))
    normals[:6] = np.array([[0, 0, 1], [0, 0, -1], [1, 0, 0], [0, 1, 0], [0, -1, 0], [-1, 0, 0]], dtype=np.float32)
    words = rng_words(n, seed)
    found = []
    k = 0
    while len(found) < n_retry:
        cand = rng_words(1 << 22, seed + 1000 + k); k += 1
        o = pcg32_outputs(cand, 2)
        r2 = o[:, 1].astype(np.float32) / np.float32(4294967296.0)
        sel = np.sqrt(np.float32(1.0) - np.minimum(r2, np.float32(0.99999994))) <= np.float32(0.001)
        found.extend(cand[sel])
    words = np.concatenate([words, np.array(found[:n_retry], dtype=np.uint32)], axis=0)
    return np.ascontiguousarray(np.concatenate([f2u(normals), words], axis=1))


def oracle_coshemi(lib, words):
    fl = u2f(words[:, :3])
    out = np.zeros((len(words), 6), dtype=np.uint32)
    st = ol.Stats()
    pdf = C.c_float()
    for i in range(len(words)):
        rng = ol.Rng(int(words[i, 3]) | (int(words[i, 4]) << 32), int(words[i, 5]) | (int(words[i, 6]) << 32))
        w = lib.orc_rand_coshemi(C.byref(rng), C.byref(pdf))
        w = lib.orc_get_rotated_to(w, ol.V3(*fl[i]))
        out[i, :3] = f2u([w.x, w.y, w.z]); out[i, 3] = f2u([pdf.value])[0]
        out[i, 4] = rng.state & 0xFFFFFFFF; out[i, 5] = rng.state >> 32
    return out


def coshemi_draws(words):
    """draws consumed per row according to the final state is awkward to invert; count retries directly"""
    o = pcg32_outputs(words[:, 3:7], 2)
    r2 = o[:, 1].astype(np.float32) / np.float32(4294967296.0)
    return np.sqrt(np.float32(1.0) - np.minimum(r2, np.float32(0.99999994))) <= np.float32(0.001)


# ------------------------------------------------------------------------------------------ trace ----
def trace_inputs(quads, seed=5, n_random=3000):
    """Rays against a scene's quads (list of [4][3] vertex arrays): random ones, and rays aimed exactly at
    vertices, edge midpoints and diagonal points of every quad (src/geometry.cpp:56-67: float edge value 0
    -> f64 fallback; ties between quads sharing the edge: strict '<', list order)."""
    g = np.random.default_rng(seed)
    verts = np.array([q for q in quads], dtype=np.float64)  # [nq, 4, 3]
    lo, hi = verts.reshape(-1, 3).min(0), verts.reshape(-1, 3).max(0)
    span = hi - lo
    rows = []
    o = g.uniform(lo, hi, size=(n_random, 3))
    d = unit(g.normal(size=(n_random, 3)))
    rows.append((o, d, np.full(n_random, -1)))
    targets = []
    for q in verts:
        targets += [q[0], q[1], q[2], q[3], (q[0] + q[1]) / 2, (q[1] + q[2]) / 2, (q[2] + q[3]) / 2, (q[3] + q[0]) / 2, (q[0] + q[2]) / 2,
                    0.25 * q[0] + 0.75 * q[2]]
    targets = np.array(targets)
    for _ in range(6):
        o = g.uniform(lo + 0.25 * span, hi - 0.25 * span, size=(len(targets), 3))
        rows.append((o, unit(targets - o), np.full(len(targets), -1)))
    # axis-parallel rays through vertices (edge functions exactly 0 in shear space)
    for axis in range(3):
        e = np.zeros(3); e[axis] = 1.0
        o = targets - e * (span[axis] * 0.37 + 0.1)
        rows.append((o, np.broadcast_to(e.astype(np.float32), o.shape), np.full(len(targets), -1)))
        rows.append((targets + e * (span[axis] * 0.41 + 0.1), np.broadcast_to((-e).astype(np.float32), o.shape), np.full(len(targets), -1)))
    # rays leaving a quad (ignore = that quad), as every bounce and shadow ray does
    nq = len(verts)
    qi = g.integers(0, nq, size=n_random)
    uv = g.uniform(0, 1, size=(n_random, 2))
    p = (verts[qi, 0] * ((1 - uv[:, :1]) * (1 - uv[:, 1:])) + verts[qi, 1] * (uv[:, :1] * (1 - uv[:, 1:])) +
         verts[qi, 2] * (uv[:, :1] * uv[:, 1:]) + verts[qi, 3] * ((1 - uv[:, :1]) * uv[:, 1:]))
    rows.append((p, unit(g.normal(size=(n_random, 3))), qi))
    O = np.concatenate([r[0] for r in rows]).astype(np.float32)
    D = np.concatenate([np.asarray(r[1], dtype=np.float32) for r in rows]).astype(np.float32)
    I = np.concatenate([r[2] for r in rows]).astype(np.int32)
    return np.ascontiguousarray(np.concatenate([f2u(O), f2u(D), I.view(np.uint32)[:, None]], axis=1))


def oracle_trace(orc, words):
    lib = orc.lib
    fl = u2f(words[:, :6])
    ign = words[:, 6].view(np.int32)
    out = np.zeros((len(words), 5), dtype=np.uint32)
    st = ol.Stats()
    hit = ol.Hit()
    for i in range(len(words)):
        ray = ol.Ray(ol.V3(*fl[i, :3]), ol.V3(*fl[i, 3:6]))
        got = lib.orc_scene_intersect(orc.scene, C.byref(ray), C.byref(hit), int(ign[i]), C.byref(st))
        if got:
            out[i, 0] = hit.prim
            out[i, 2] = f2u([hit.dist])[0]; out[i, 3:5] = f2u([hit.st.x, hit.st.y])
        else:
            out[i, 0] = 0xFFFFFFFF; out[i, 2] = f2u([np.inf])[0]
    return out, st


# ------------------------------------------------------------------------------------ rand_choice ----
def rand_choice_inputs(seed=6, n=4000):
    """stream + n; the large n make Lemire's redraw (bits/uniform_int_dist.h) common (probability
    (2^32 mod n)/2^32: ~1/4 for n = 3*2^30, ~1/2 for 2^31 + 1)."""
    ns = np.array([1, 2, 3, 6, 19, 255, 1000003, 3 << 30, (1 << 31) + 1, (1 << 32) - 1], dtype=np.uint32)
    w = rng_words(n, seed)
    return np.ascontiguousarray(np.concatenate([w, ns[np.arange(n) % len(ns)][:, None]], axis=1))


def oracle_rand_choice(lib, words):
    out = np.zeros((len(words), 3), dtype=np.uint32)
    st = ol.Stats()
    for i in range(len(words)):
        rng = ol.Rng(int(words[i, 0]) | (int(words[i, 1]) << 32), int(words[i, 2]) | (int(words[i, 3]) << 32))
        out[i, 0] = lib.orc_rand_choice(C.byref(rng), int(words[i, 4]))
        out[i, 1] = rng.state & 0xFFFFFFFF; out[i, 2] = rng.state >> 32
    return out


def lemire_redraws(words):
    """rows whose first draw triggers a redraw"""
    o = pcg32_outputs(words[:, :4], 1)[:, 0].astype(np.uint64)
    n = words[:, 4].astype(np.uint64)
    low = (o * n) & np.uint64(0xFFFFFFFF)
    thr = (np.uint64(1 << 32) - n) % n
    return (low < n) & (low < thr)


# ------------------------------------------------------------------------------------------ fmath ----
def fmath_inputs(seed=7, n=20000):
    g = np.random.default_rng(seed)
    x = np.concatenate([g.uniform(-1, 1, n), g.uniform(-7, 7, n), g.uniform(-1, 1, n // 4) * 1e-6, 1 - g.uniform(0, 1, n // 4) * 1e-5,
                        -1 + g.uniform(0, 1, n // 4) * 1e-5, g.uniform(-1e5, 1e5, n // 4),
                        [0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 1.0000001, np.pi, np.pi / 2, 2 * np.pi, 1048576.0, 1048577.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45]])
    return f2u(x.astype(np.float32))[:, None]


def oracle_fmath(lib, words):
    x = u2f(words[:, 0])
    out = np.zeros((len(x), 5), dtype=np.float32)
    for i, v in enumerate(x):
        c = C.c_float(float(v))
        s_, c_, a_ = lib.orc_sinf(c), lib.orc_cosf(c), lib.orc_acosf(c)
        out[i] = (s_, c_, a_, s_, c_)
    return out


# ------------------------------------------------------------------------------- flux -> XYZ, rand_1f ----
def flux_inputs(lambda_min, lambda_step, seed=8, n=3000):
    g = np.random.default_rng(seed)
    flux = g.uniform(0, 50, size=(n, 4)).astype(np.float32)
    lam = (lambda_min + g.uniform(0, 1, size=(n, 1)) * lambda_step).astype(np.float32)
    lam[:4, 0] = (lambda_min, lambda_min + lambda_step, np.nextafter(np.float32(lambda_min + lambda_step), np.float32(0)), lambda_min + 0.5 * lambda_step)
    return np.ascontiguousarray(np.concatenate([f2u(flux), f2u(lam)], axis=1))


def oracle_flux(orc, words):
    fl = u2f(words)
    out = np.zeros((len(words), 3), dtype=np.float32)
    xyz = (C.c_float * 3)()
    for i in range(len(words)):
        flux = (C.c_float * 4)(*[float(v) for v in fl[i, :4]])
        orc.lib.orc_specradflux_to_ciexyz_hero(orc.color, flux, C.c_float(float(fl[i, 4])), xyz)
        out[i] = xyz[:]
    return out


def oracle_rand_1f(lib, words):
    out = np.zeros((len(words), 3), dtype=np.uint32)
    for i in range(len(words)):
        rng = ol.Rng(int(words[i, 0]) | (int(words[i, 1]) << 32), int(words[i, 2]) | (int(words[i, 3]) << 32))
        out[i, 0] = f2u([lib.orc_rand_1f(C.byref(rng))])[0]
        out[i, 1] = rng.state & 0xFFFFFFFF; out[i, 2] = rng.state >> 32
    return out
