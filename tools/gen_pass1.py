#!/usr/bin/env python3
"""Generates simple_spectral_amd/csrc/ssx_pass1_gen.h: pass 1 of the intersection (the wave-uniform
edge-function filter of trace(), csrc/ssx_kernels.hip) written out straight-line for the mesh TOPOLOGIES of
the reference's built-in scenes (src/scene.cpp:32-415: the Cornell box of "cornell" / "cornell-srgb", and
"plane-srgb").

Why: the generic pass 1 treats every quad alone -- 4 vertex shears and 5 edge functions per quad.  The Cornell
box's 19 quads have 76 vertex slots but 28 distinct vertices, and 95 edge slots but 63 distinct edges (walls,
floor and ceiling share corners, a block's faces share edges); an edge function is antisymmetric in its two
vertices, E(p,q) = p.y*q.x - p.x*q.y = -E(q,p) exactly, so a shared edge is evaluated once.  Sharing needs
every sheared vertex in a register of its own, i.e. code with static indices: registers cannot be indexed by
a wave-uniform index cheaper than re-shearing.  So the sharing structure -- which quad corners coincide, and
nothing else; positions stay run-time data -- is compiled in: one straight-line function per topology.  A
scene whose corners coincide in another pattern (tests/crafted.py, any user scene) runs the generic loop.

Every value the filter tests is the same float as in the generic loop (same operands, same operations: a
shared vertex is sheared from the same coordinates, a shared edge is the same two products), so the candidate
sets are identical and results do not depend on which variant runs.

The header is committed; tests/test_host_and_abi.py regenerates it and checks that it is up to date with
the host's scene builder.      usage: python tools/gen_pass1.py [--check]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "simple_spectral_amd", "csrc", "ssx_pass1_gen.h")

# (name, id, host scene that has it, cull triangles behind the ray's origin in pass 1)
# The cull (plane topology only): a triangle whose three scaled depths Sz * (v - o)[kz] -- the reference's ABCz (src/geometry.cpp:75) -- are all
# <= 0 has T = U*Az + V*Bz + W*Cz of the sign opposite to det = U + V + W (the edge values of a candidate share a sign), or T = +-0: the reference
# rejects it at the sign test (:80-83) or at dist >= EPS (:88), it never becomes the hit and never stands in for its quad's other triangle.  Such a
# triangle need not be a candidate: its flag is ORed with the sign bit of max3(Sz*zA, Sz*zB, Sz*zC) (set iff all three are < 0 or -0; a NaN ray
# hits nothing either way).  Costs a multiply per vertex and two instructions per triangle in pass 1 (all lanes busy) and saves a pass-2 trip
# per trace in plane-srgb, where every line through the box meets a wall behind its origin (camera rays 3 -> 2 candidates, bounce rays 2 -> 1).
# In the Cornell box the candidates behind the origin are too few for the 104 instructions the test would add to its pass 1.
TOPOLOGIES = [("cornell", 1, "cornell", False), ("plane", 2, "plane-srgb", True)]


def scene_vids(scene_name):
    """[[vid x 4] per quad]: corners numbered by first occurrence of their position (bitwise float equality)."""
    import numpy as np
    from simple_spectral_amd.renderer import Scene
    scene = Scene(scene_name, texture="test-img.png" if scene_name != "cornell" else None)  # keeps the description alive
    d = scene.desc.contents
    ids, vids = {}, []
    for q in range(d.n_quads):
        row = []
        for v in (d.quads[q].v00, d.quads[q].v10, d.quads[q].v11, d.quads[q].v01):
            key = np.array(v.pos[:], dtype=np.float32).tobytes()
            row.append(ids.setdefault(key, len(ids)))
        vids.append(row)
    scene.close()
    return vids


def components(vids):
    """Runs of consecutive quads [(first_quad, n_quads, [vertex ids])]: the connected components of the mesh
    (quads connected through shared vertices) where those are contiguous in list order, else maximal contiguous
    pieces of them (a vertex shared between two runs is then sheared in both: still the same float).  The runs
    bound register pressure -- a run's vertices die with it -- and list order must be kept because the flags are
    shifted in in triangle order."""
    n = len(vids)
    parent = list(range(n))
    def find(i):
        while parent[i] != i:
            parent[i] = parent[parent[i]]; i = parent[i]
        return i
    owner = {}
    for q, row in enumerate(vids):
        for v in row:
            if v in owner:
                parent[find(q)] = find(owner[v])
            else:
                owner[v] = q
    comps = []
    q = 0
    while q < n:
        end = q + 1
        while end < n and find(end) == find(q):
            end += 1
        verts = sorted(set(v for row in vids[q:end] for v in row))
        comps.append((q, end - q, verts))
        q = end
    return comps


def E(u, v):
    """source text of the edge function E(u, v) through the one stored orientation"""
    if u == v:
        return "0.0f"
    return "e%d_%d" % (u, v) if u < v else "-e%d_%d" % (v, u)


def emit_topology(name, vids, cull_behind=False):
    nq = len(vids)
    lines = []
    lines.append("// topology \"%s\": %d quads, %d distinct vertices of %d corners" % (name, nq, 1 + max(max(r) for r in vids), 4 * nq))
    nv = 1 + max(max(r) for r in vids)
    lines.append("// vt: the distinct-vertex table of the ray's axis permutation: {x,y} pairs of all vertices, then their z (ssx_blob.h)")
    lines.append("__device__ __forceinline__ void pass1_%s(const float* vt, const RaySetup& rs, uint32_t& acc0, uint32_t& acc1) {" % name)
    lines.append("\tconst float* vz = vt + %d;" % (2 * nv))
    n_edges = 0
    for first, count, verts in components(vids):
        lines.append("\t{ // quads %d..%d" % (first, first + count - 1))
        for k in verts:
            lines.append("\t\tconst float z%d = vz[%d] - rs.okz, x%d = (vt[%d] - rs.okx) - rs.Sx * z%d, y%d = (vt[%d] - rs.oky) - rs.Sy * z%d;"
                         % (k, k, k, 2 * k, k, k, 2 * k + 1, k))
            if cull_behind:
                lines.append("\t\tconst float b%d = rs.Sz * z%d;" % (k, k))
        edges = set()
        for q in range(first, first + count):
            a, b, c, d = vids[q]
            for u, v in ((b, c), (c, a), (a, b), (c, d), (d, a)):
                if u != v:
                    edges.add((min(u, v), max(u, v)))
        for u, v in sorted(edges):
            lines.append("\t\tconst float e%d_%d = y%d * x%d - x%d * y%d;" % (u, v, u, v, u, v))
        n_edges += len(edges)
        for q in range(first, first + count):
            a, b, c, d = vids[q]
            acc = "acc0" if q < 16 else "acc1"
            # tri0 = (a,b,c): U = E(b,c), V = E(c,a), W = E(a,b);  tri1 = (a,c,d): U = E(c,d), V = E(d,a), W = E(a,c)
            for tri, vs3 in (((E(b, c), E(c, a), E(a, b)), (a, b, c)), ((E(c, d), E(d, a), E(a, c)), (a, c, d))):
                behind = " | SSX_P1_BEHIND(b%d, b%d, b%d)" % vs3 if cull_behind else ""
                lines.append("\t\t{ const float u = %s, v = %s, w = %s; %s = __builtin_amdgcn_alignbit(%s, __float_as_uint(__builtin_fmaf(__builtin_fminf(__builtin_fminf(u, v), w), __builtin_fmaxf(__builtin_fmaxf(u, v), w), 0.0f))%s, 31u); } // quad %d"
                             % (tri[0], tri[1], tri[2], acc, acc, behind, q))
        lines.append("\t}")
    lines.append("}")
    lines[0] += ", %d distinct edges of %d" % (n_edges, 5 * nq)
    return lines


def generate():
    out = ["// ssx_pass1_gen.h -- GENERATED by tools/gen_pass1.py (see there for the why); do not edit.",
           "// Pass 1 of trace() straight-line for the mesh topologies of the reference's built-in scenes.",
           "#pragma once",
           "// sign bit set iff the three scaled depths are all < 0 (or -0): the triangle lies behind the ray's origin and is no candidate (tools/gen_pass1.py)",
           "#ifdef SSX_NO_BEHIND_CULL // (A/B builds)",
           "#define SSX_P1_BEHIND(a, b, c) 0u",
           "#else",
           "#define SSX_P1_BEHIND(a, b, c) __float_as_uint(__builtin_fmaxf(__builtin_fmaxf((a), (b)), (c)))",
           "#endif", ""]
    host = ["// corner -> distinct-vertex id per quad (v00, v10, v11, v01), numbered by first occurrence: what ssx_upload_scene",
            "// compares an uploaded scene's sharing pattern with, and what pass 2 of the specialised kernels looks vertices up by",
            "struct SsxTopology { uint32_t id, n_quads, n_verts; const uint8_t (*vid)[4]; };"]
    dev = []
    table = []
    for name, tid, scene, cull in TOPOLOGIES:
        vids = scene_vids(scene)
        nv = 1 + max(max(r) for r in vids)
        host.append("static const uint8_t ssx_topo_%s_vid[%d][4] = { %s };" % (name, len(vids), ", ".join("{ %d, %d, %d, %d }" % tuple(r) for r in vids)))
        table.append("{ %du, %du, %du, ssx_topo_%s_vid }" % (tid, len(vids), nv, name))
        dev += emit_topology(name, vids, cull) + [""]
    host.append("static const SsxTopology ssx_topologies[%d] = { %s };" % (len(table), ", ".join(table)))
    return "\n".join(out + host + [""] + dev)


if __name__ == "__main__":
    text = generate()
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(OUT) and open(OUT).read() == text else 1)
    open(OUT, "w").write(text)
    print("wrote", OUT, "(%d lines)" % text.count("\n"))
