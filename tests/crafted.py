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
