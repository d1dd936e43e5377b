"""Crafted scenes that force the rare branches of the integrator (reference file:line in each
builder), for both sides of the parity tests.  TEST INFRASTRUCTURE (uses tests/custom_scene.py)."""
import numpy as np

import custom_scene as cs

# materials of the base scene "cornell" (oracle_scene.c / src/scene.cpp:66-110)
WHITE, GREEN, RED, LIGHT = 0, 3, 4, 5


def _room(c, lo=-4.0, hi=4.0, mat=WHITE):
    """closed axis-aligned box, normals inward (vertex order as the reference's plane-srgb box, src/scene.cpp:372-409)"""
    l, h = lo, hi
    c.add_quad((l, l, h), (l, l, l), (l, h, l), (l, h, h), mat)
    c.add_quad((h, l, l), (h, l, h), (h, h, h), (h, h, l), GREEN)
    c.add_quad((l, l, h), (h, l, h), (h, l, l), (l, l, l), mat)
    c.add_quad((h, h, h), (l, h, h), (l, h, l), (h, h, l), mat)
    c.add_quad((l, l, l), (h, l, l), (h, h, l), (l, h, l), RED)
    c.add_quad((h, l, h), (l, l, h), (l, h, h), (h, h, h), mat)


def degenerate_light_scene(view="edge_ab"):
    """SphericalTriangle's degenerate ladder (src/util/spherical-tri.cpp:74-123), zero-area light
    triangles (pdf = 1/0, src/geometry.cpp:115) and a light too small to subtend an angle.

    A unit light quad hangs at y = 1.  Small receiver quads sit ON THE EXTENSION of two of its
    edges, so that from every point of them two of the three directions toward a light triangle's
    vertices coincide (their dot product rounds to 1.0f, the side is 0, its sine is 0):
      view "edge_ab": receiver beyond v10 on the line v00-v10 -> side c of tri0 is 0 (:85-92: alpha = pi/2)
      view "edge_bc": receiver beyond v11 on the line v10-v11 -> side a of tri0 is 0 (:105-113: alpha = acos(.))
      view "far":     a second, 1e-4-sized light seen from metres away -> all sides 0 (:114-122: NaN)
      view "room":    the whole room (every branch mixed with regular paths)
    The second light's tri0 is collapsed (v10 == v11): zero area, NaN normal, can never be hit, but is
    sampled with probability 1/4."""
    c = cs.CustomScene("cornell", keep_quads=False)
    _room(c)
    c.add_quad((0, 1, 0), (1, 1, 0), (1, 1, 1), (0, 1, 1), LIGHT)                      # quad 6: the light, facing down
    e = 2e-3
    c.add_quad((3, 1 - e, -e), (3, 1 - e, e), (3, 1 + e, e), (3, 1 + e, -e), WHITE)    # 7: on the line through v00-v10 (x axis at y=1, z=0)
    c.add_quad((1 - e, 1 - e, 3), (1 + e, 1 - e, 3), (1 + e, 1 + e, 3), (1 - e, 1 + e, 3), WHITE)  # 8: on the line through v10-v11
    t = 1e-4
    c.add_quad((-2, 2, -2), (-2 + t, 2, -2), (-2 + t, 2, -2), (-2, 2, -2 + t), LIGHT)  # 9: tiny light, tri0 collapsed
    if view == "edge_ab":
        c.set_camera((2.0, 1.0, 0.0), (3.0, 1.0, 0.0), up=(0, 1, 0), vfov_deg=0.2)
    elif view == "edge_bc":
        c.set_camera((1.0, 1.0, 2.0), (1.0, 1.0, 3.0), up=(0, 1, 0), vfov_deg=0.2)
    elif view == "far":
        c.set_camera((3.0, -3.0, 3.0), (3.9, -3.9, 3.9), up=(0, 1, 0), vfov_deg=30.0)
    else:
        c.set_camera((3.5, -3.0, -3.5), (0.0, 0.5, 0.5), up=(0, 1, 0), vfov_deg=60.0)
    return c


def shared_edge_scene():
    """Rays through shared vertices and edges: the f64 fallback of the watertight test
    (src/geometry.cpp:56-67) and its tie rules (strict '<', first quad in list order wins,
    src/scene.cpp:433-445; tri0 before tri1, src/geometry.cpp:128-139).  Four coplanar quads meet in
    one vertex on the camera axis; the field of view is a few float ulps wide, so most camera rays
    have an edge function that is exactly 0 in float."""
    c = cs.CustomScene("cornell", keep_quads=False)
    _room(c)
    c.add_quad((-1, 3.9, -1), (1, 3.9, -1), (1, 3.9, 1), (-1, 3.9, 1), LIGHT)
    z = 1.0
    c.add_quad((0, 0, z), (1, 0, z), (1, 1, z), (0, 1, z), WHITE)
    c.add_quad((-1, 0, z), (0, 0, z), (0, 1, z), (-1, 1, z), GREEN)
    c.add_quad((-1, -1, z), (0, -1, z), (0, 0, z), (-1, 0, z), RED)
    c.add_quad((0, -1, z), (1, -1, z), (1, 0, z), (0, 0, z), WHITE)
    c.set_camera((0.0, 0.0, -2.0), (0.0, 0.0, 1.0), up=(0, 1, 0), vfov_deg=2e-5)
    return c


def triangle_scene():
    """Top-level PrimTri primitives (src/geometry.hpp:55-74, src/scene.hpp:39-41): one as a LIGHT -- sampled by
    PrimTri::get_rand_toward (src/geometry.cpp:103-116): no triangle pick, so one random number fewer than a quad
    light, and no halving of the pdf (:141-145) -- next to a quad light (the light pick, src/scene.cpp:417-431,
    then mixes both kinds), two as occluders (one triangle each: where a quad's second triangle would be, rays pass)."""
    c = cs.CustomScene("cornell", keep_quads=False)
    _room(c)
    c.add_tri((-3, 3.9, -3), (-1, 3.9, -3), (-3, 3.9, -1), LIGHT)                    # 6: triangle light under the ceiling
    c.add_quad((1, 3.9, 1), (2.5, 3.9, 1), (2.5, 3.9, 2.5), (1, 3.9, 2.5), LIGHT)    # 7: quad light
    c.add_tri((-2.5, 0.5, 0.5), (0.5, 0.5, 0.0), (-1.0, 0.5, 2.5), GREEN)             # 8: a triangle hovering over the floor
    c.add_tri((0.0, -1.0, -1.0), (2.0, -1.5, 0.0), (1.0, 1.5, 0.5), RED)              # 9: a tilted one
    c.add_quad((-3.5, -4 + 1e-3, -3.5), (-1.5, -4 + 1e-3, -3.5), (-1.5, -4 + 1e-3, -1.5), (-3.5, -4 + 1e-3, -1.5), WHITE)  # 10: a quad after the triangles
    c.set_camera((3.4, -2.5, -3.4), (-0.5, 0.3, 0.5), up=(0, 1, 0), vfov_deg=65.0)
    return c


def many_prims_scene(n_prims, observer=1931, seed=9):
    """More than 32 primitives (the reference's Scene::intersect loops over any number, src/scene.cpp:433-445): the room, then
    a cloud of small quads and triangles in all orientations, lights among them at list positions in different groups of
    32, the last primitive a light too.  n_prims = 70 with the CIE 1931 tables keeps every table in LDS; 128 with the
    CIE 2006 tables overflows it: the permuted vertex table is then read from HBM."""
    g = np.random.default_rng(seed)
    c = cs.CustomScene("cornell", observer=observer, keep_quads=False)
    _room(c)
    mats = (WHITE, GREEN, RED)
    light_at = {6, 20, 33, 64, 97, n_prims - 1}
    while len(c.quads) < n_prims:
        i = len(c.quads)
        ctr = g.uniform(-3.2, 3.2, size=3)
        u = g.normal(size=3); u /= np.linalg.norm(u)
        v = np.cross(u, g.normal(size=3)); v /= np.linalg.norm(v)
        su, sv = g.uniform(0.25, 0.9, size=2)
        m = LIGHT if i in light_at else mats[i % 3]
        p00, p10, p11, p01 = ctr - su * u - sv * v, ctr + su * u - sv * v, ctr + su * u + sv * v, ctr - su * u + sv * v
        if i % 5 == 3:
            c.add_tri(p00, p10, p11, m)
        else:
            c.add_quad(p00, p10, p11, p01, m)
    c.set_camera((3.6, 0.2, -3.6), (0.0, 0.0, 0.0), up=(0, 1, 0), vfov_deg=70.0)
    return c


def many_textures_scene(n_textures=9, seed=4):
    """The Cornell box with every wall, the floor and the faces of the short block on textures of their own (different sizes,
    incl. 1 x 1 and non-square): more textures than the reference's scenes use (src/scene.cpp: one); MaterialLambertian holds
    any texture per material (src/material.cpp:10-29)."""
    g = np.random.default_rng(seed)
    c = cs.CustomScene("cornell-srgb")
    sizes = [(1, 1), (2, 3), (7, 5), (16, 16), (33, 9), (64, 48), (5, 128), (31, 31), (256, 2), (12, 12), (3, 3), (8, 1)]
    targets = [0, 6, 7, 9, 10, 11, 12, 13, 2, 3, 4, 5]
    for k in range(n_textures - len(c.textures)):
        h, w = sizes[k % len(sizes)]
        c.textures.append(g.integers(0, 256, size=(h, w, 3), dtype=np.uint8))
        if k < len(targets): # (textures beyond the quads at hand are uploaded and unused)
            m = c.add_material(albedo_texture=len(c.textures) - 1)
            pos, st, _ = c.quads[targets[k]]
            c.quads[targets[k]] = (pos, st, m)
    return c


def random_scene(seed):
    """A scene drawn from `seed`: 3..40 (one in six: up to 115) quads and triangles of all sizes and orientations, sometimes inside a closed room, with
    Lambertian (constant / textured), mirror and emissive materials, at least one light (quad or triangle), a random camera.
    Returns (scene, render options): indirect_only / explicit light sampling / flat-field correction are drawn as well.  The
    differential fuzz of the parity tests: combinations no hand-made scene has (a mirror next to a triangle light in a build
    without explicit light sampling, textures on triangles, lights seen edge-on, ...)."""
    g = np.random.default_rng(1000 + seed)
    c = cs.CustomScene("cornell-srgb", keep_quads=False, observer=(1931, 2006)[int(g.integers(0, 4) == 0)])
    mats = [WHITE, GREEN, RED]
    white_spectrum = c.materials[WHITE]["albedo_spectrum"]
    mats.append(c.add_material(kind=1, albedo_spectrum=white_spectrum))                 # a mirror (src/material.cpp:146-167)
    for _ in range(int(g.integers(0, 3))):                                              # up to two more textures
        h, w = (int(v) for v in g.integers(1, 24, size=2))
        c.textures.append(g.integers(0, 256, size=(h, w, 3), dtype=np.uint8))
    for t in range(len(c.textures)):
        mats.append(c.add_material(albedo_texture=t))
    if g.integers(0, 2):
        _room(c, mat=mats[int(g.integers(0, len(mats)))])
    n = int(g.integers(3, 41)) if g.integers(0, 6) else int(g.integers(41, 110))      # one scene in six: several groups of 32 primitives
    n_lights = 0
    while len(c.quads) < n or n_lights == 0:
        ctr = g.uniform(-3.0, 3.0, size=3)
        u = g.normal(size=3); u /= np.linalg.norm(u)
        v = np.cross(u, g.normal(size=3)); v /= np.linalg.norm(v)
        su, sv = np.exp(g.uniform(np.log(0.05), np.log(2.5), size=2))
        light = g.integers(0, 6) == 0 or (len(c.quads) >= n and n_lights == 0)
        m = LIGHT if light else mats[int(g.integers(0, len(mats)))]
        n_lights += int(light)
        skew = g.uniform(-0.3, 0.3) * su * u if g.integers(0, 3) == 0 else 0.0 * u       # some quads are not rectangles
        p00, p10, p11, p01 = ctr - su * u - sv * v, ctr + su * u - sv * v, ctr + su * u + sv * v + skew, ctr - su * u + sv * v
        if g.integers(0, 4) == 0:
            c.add_tri(p00, p10, p11, m)
        else:
            c.add_quad(p00, p10, p11, p01, m)
    eye = g.uniform(-3.8, 3.8, size=3)
    c.set_camera(tuple(eye), tuple(g.uniform(-1.0, 1.0, size=3)), up=(0, 1, 0), vfov_deg=float(g.uniform(20.0, 90.0)))
    opts = dict(indirect_only=bool(g.integers(0, 4) == 0), els=bool(g.integers(0, 3) != 0), flat_field=bool(g.integers(0, 4) != 0))
    return c, opts


def origin_light_scene():
    """A light with a vertex exactly at the origin: shading points a denormal-ish distance from it give squared lengths below
    2^-100, outside the proven domain of the kernel's fast exact sqrt (csrc/ssx_exact.h) -- the kernel must take the plain IEEE
    path there (normalize3_any, func_bar) and still equal the reference's glm::normalize."""
    c = cs.CustomScene("cornell", keep_quads=False)
    _room(c)
    c.add_quad((0, 0, 0), (1, 0, 0), (1, 0, 1), (0, 0, 1), LIGHT)
    c.set_camera((2.0, 2.0, -3.0), (0.5, 0.0, 0.5), up=(0, 1, 0), vfov_deg=40.0)
    return c


def _distinct_corners(c):
    """{position bits: position} over the corners of the scene's quads, in order of first occurrence"""
    pts = {}
    for pos, _, _ in c.quads:
        for p in pos:
            pts.setdefault(np.asarray(p, dtype=np.float32).tobytes(), np.asarray(p, dtype=np.float64))
    return pts


def warped_builtin(seed):
    """One of the reference's built-in scenes (src/scene.cpp:32-415) with every distinct corner moved -- the same way wherever it occurs, so the
    corners still coincide in the built-in pattern and the library keeps the kernel whose pass 1 is specialised to that mesh topology
    (csrc/ssx_pass1_gen.h: the sharing pattern is compiled in, positions are run-time data).  The random scenes of `random_scene` run the generic
    kernel or one compiled at upload; this is the fuzz of the kernels the headline numbers are measured on: the Cornell box and the plane scene
    rotated, scaled by 2^-10 .. 2^10 and translated (the light stays in the ceiling's plane: shadow rays inside the plane of ten triangles, now
    along no axis), with every corner jittered (quads that are neither planar nor rectangles, lights included, rooms that leak), flattened into
    slivers, or sheared -- with a camera that follows.  Returns (scene, name of the base scene, render options)."""
    g = np.random.default_rng(5000 + seed)
    base = ("cornell-srgb", "plane-srgb", "cornell")[seed % 3]
    kind = (seed // 3) % 4            # 0: similarity, 1: jitter, 2: similarity + jitter, 3: flatten + shear
    c = cs.CustomScene(base, observer=(1931, 2006)[int(g.integers(0, 4) == 0)])
    pts = _distinct_corners(c)
    allp = np.array(list(pts.values()))
    ctr = 0.5 * (allp.min(axis=0) + allp.max(axis=0))
    size = float(np.abs(allp - ctr).max())
    q, _ = np.linalg.qr(g.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    if kind in (0, 2):
        s = 2.0 ** float(g.integers(-10, 11))
        A = s * q
        new_ctr = ctr * s + g.uniform(-1.0, 1.0, size=3) * size * s
    elif kind == 3:
        flat = np.diag([1.0, 1.0, 2.0 ** float(-g.integers(3, 13))])           # one axis of a rotated frame pressed flat: every triangle a sliver from most directions
        shear = np.eye(3); shear[0, 1] = g.uniform(-0.8, 0.8)
        A = q @ flat @ shear @ q.T
        new_ctr = ctr
    else:
        A = np.eye(3)
        new_ctr = ctr
    amp = 0.03 * size if kind in (1, 2) else 0.0
    while True:
        moved = {k: (A @ ((p + g.normal(size=3) * amp) - ctr) + new_ctr).astype(np.float32) for k, p in pts.items()}
        if len({m.tobytes() for m in moved.values()}) == len(moved):               # distinct corners stay distinct (else the sharing pattern would change)
            break
    for i, (pos, st, m) in enumerate(c.quads):
        c.quads[i] = (np.array([moved[np.asarray(p, dtype=np.float32).tobytes()] for p in pos], dtype=np.float32), st, m)
    eye = A @ (c.cam_pos.astype(np.float64) - ctr) + new_ctr
    target = new_ctr + A @ (g.uniform(-0.2, 0.2, size=3) * size)
    if kind == 3 and base != "plane-srgb":                                         # (the flattened box seen from its opening is a line: look around from inside)
        eye = new_ctr + A @ (g.uniform(-0.35, 0.35, size=3) * size)
        target = new_ctr + A @ (g.uniform(-0.35, 0.35, size=3) * size)
    up = A @ np.array([0.0, 1.0, 0.0]); up /= np.linalg.norm(up)
    c.set_camera(tuple(eye), tuple(target), up=tuple(up), vfov_deg=float(g.uniform(30.0, 70.0)))
    opts = dict(indirect_only=bool(g.integers(0, 4) == 0), els=bool(g.integers(0, 4) != 0), flat_field=bool(g.integers(0, 4) != 0))
    return c, base, opts
